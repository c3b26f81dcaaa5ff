# Developer aid: core microbenchmark over the variants in build/ab (names as arguments)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$v.so timeout -k 10 120 python tools/bench_core.py R1_2 2>&1 | tail -1 || exit 1
done
