cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do echo "$v stamps:"; RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$v.so timeout -k 10 200 python tools/exp_demod_stamps.py 2>&1 | tail -3; done
echo "tree stamps:"; timeout -k 10 200 python tools/exp_demod_stamps.py 2>&1 | tail -3
