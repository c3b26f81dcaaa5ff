cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modes.py -m gpu -x -q -k "demod or rx_fused or bench_workload or thousands or other_modes or offsets or burst or adaptor or cox or edge" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 200 python tools/exp_demod_stamps.py 2>&1 | tail -4
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/b.log 2>&1
tail -1 $O/b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['frames_decoded_last_step'])"
