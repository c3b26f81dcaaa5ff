# Developer aid: alternating bench.py runs of build/ab/<A>.so and build/ab/<B>.so (+ env of B in $3), then the core microbenchmark
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['frames_decoded_last_step'], d['config']['frames_bytes_equal_tx_last_step'])"; }
for rep in 1 2; do
  echo "$1:"; RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$1.so timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline | tail -1 | line || exit 1
  echo "$2 ($3):"; env $3 RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$2.so timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline | tail -1 | line || exit 1
done
RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$1.so timeout -k 10 120 python tools/bench_core.py R1_2 2>&1 | tail -1 && env $3 RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$2.so timeout -k 10 120 python tools/bench_core.py R1_2 2>&1 | tail -1
