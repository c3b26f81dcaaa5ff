cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for r in R1_4 R1_3 R2_3 R3_4 R5_6; do
  timeout -k 10 120 python tools/bench_core.py $r 98304 2>&1 | tail -1
  RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$1.so timeout -k 10 120 python tools/bench_core.py $r 98304 2>&1 | tail -1
done
