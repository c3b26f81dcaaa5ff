cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02v; mkdir -p $O
for c in 0 2 4 8; do
  RIA_CTL_CUS=$c timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/b_$c.log 2>&1
  tail -1 $O/b_$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ctl cus $c', d['value'], d['ms_per_step'], d['config']['frames_decoded_last_step'])"
done
RIA_CTL_CUS=4 timeout -k 10 600 python -m pytest tests/test_gpu_modes.py -m gpu -x -q -k "split_modes or offsets" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
RIA_CTL_CUS=4 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --steps-only --steps 4 --warmup 1 > $O/st.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02v/st/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f))):
    if any(k in r['Name'] for k in ('mark','chain','finalize','validate','recovery_list','cascade','phase0')):
        print(r['Name'][:58].ljust(60), r['Calls'], 'avg %.3f min %.3f max %.3f'%(float(r['AverageNs'])/1e6,float(r['MinNs'])/1e6,float(r['MaxNs'])/1e6))
PY
