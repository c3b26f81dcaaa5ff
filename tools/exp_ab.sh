# Developer aid: alternating bench.py runs of build/ab/<A>.so and build/ab/<B>.so, then demod stamps and the core microbenchmark
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['frames_decoded_last_step'])"; }
for rep in 1 2; do for v in $1 $2; do
  echo "$v:"; RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$v.so timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | tail -1 | line || exit 1
done; done
for v in $1 $2; do echo "$v stamps:"; RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$v.so timeout -k 10 200 python tools/exp_demod_stamps.py 2>&1 | tail -3; done
for v in $1 $2; do RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$v.so timeout -k 10 120 python tools/bench_core.py R1_2 2>&1 | tail -1; done
