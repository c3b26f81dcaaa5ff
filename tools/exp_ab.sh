# Developer aid: alternating bench.py runs of the in-tree library and build/ab/<variant>.so, then demod stamps and the core microbenchmark
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=$GRAFT_REPO_ROOT/build/ab/$1.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['frames_decoded_last_step'])"; }
for rep in 1 2; do
  echo "tree:"; timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | tail -1 | line || exit 1
  echo "$1:";  RIA_GPU_LIB=$B timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | tail -1 | line || exit 1
done
echo "tree stamps:"; timeout -k 10 200 python tools/exp_demod_stamps.py 2>&1 | tail -3
echo "$1 stamps:"; RIA_GPU_LIB=$B timeout -k 10 200 python tools/exp_demod_stamps.py 2>&1 | tail -3
timeout -k 10 120 python tools/bench_core.py R1_2 2>&1 | tail -1
RIA_GPU_LIB=$B timeout -k 10 120 python tools/bench_core.py R1_2 2>&1 | tail -1
