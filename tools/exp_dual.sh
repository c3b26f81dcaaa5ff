cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['frames_decoded_last_step'])"; }
for rep in 1 2; do
  echo "single:"; timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | tail -1 | line || exit 1
  echo "dual:";  RIA_DUAL=1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | tail -1 | line || exit 1
done
