cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02t; mkdir -p $O
for p in 1 2 3; do for g in 3072 2816; do
  RIA_PERSIST_GRID=$g RIA_SPLIT_PARTS=$p timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_${g}_$p.log 2>&1
  tail -1 $O/bench_${g}_$p.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid $g parts $p', d['value'], d['ms_per_step'])"
done; done
