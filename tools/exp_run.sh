# Developer aid: A/B of two builds of the library on one box (RIA_GPU_LIB), alternating runs; then the GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-ab}; mkdir -p $O
A=$GRAFT_REPO_ROOT/build/ab/base.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['frames_decoded_last_step'])"; }
for rep in 1 2; do
  echo "base:"; RIA_GPU_LIB=$A timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | tail -1 | line || exit 1
  echo "new:";  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | tail -1 | line || exit 1
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
