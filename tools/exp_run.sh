cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02u; mkdir -p $O
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['frames_decoded_last_step'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --steps-only --steps 4 --warmup 1 > $O/st.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02u/st/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f))):
    if 'ria::' in r['Name'] and not any(k in r['Name'] for k in ('tx_frames','channel','make_frames')):
        print(r['Name'][:58].ljust(60), r['Calls'], 'avg %.3f min %.3f max %.3f'%(float(r['AverageNs'])/1e6,float(r['MinNs'])/1e6,float(r['MaxNs'])/1e6))
PY
