cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python tools/bench_sync.py 2>&1 | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(v.get('preambles_per_s') or v.get('spans_per_s') or v.get('buffers_per_s') or v.get('frames_per_s')) for k,v in d.items()})"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "mcdpsk or harq or ladder" 2>&1 | tail -2
