# Developer aid: bench.py under the split settings of ria_gpu_rx_batch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
echo "nosplit:"; RIA_NO_SPLIT=1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | tail -1 | line || exit 1
for p in 2 3 4 5 6; do echo "parts $p:"; RIA_SPLIT_PARTS=$p timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | tail -1 | line || exit 1; done
for m in 0 200000 ; do echo "bankopt $m"; RIA_BANKOPT_MOVES=$m timeout -k 10 300 python tools/bench_core.py R1_2 2>&1 | tail -1; done
timeout -k 10 300 python tools/bench_core.py R1_2 2>&1 | tail -1
