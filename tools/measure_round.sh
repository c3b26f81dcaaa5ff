#!/bin/bash
# One measurement session on the MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/measure_round.sh r01i'
# GPU test suite, smoke, the contract bench, rocprofv3 kernel trace, the PMC passes (each counter set in its own run,
# only with --kernel-trace) and the sync bench.  Outputs under gpurun_out/<tag>/; tools/make_profile_summary.py turns
# them into profiles/<tag>_*.  A step that times out ends the session (no further GPU work after a hang).
TAG=${1:-r01x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; rm -rf "$O"; mkdir -p "$O"
step() { local name=$1; shift; "$@"; local rc=$?; echo "$name rc=$rc" | tee -a "$O/progress.log"; if [ $rc -ge 124 ]; then exit $rc; fi; }
run_pytest() { timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1; }
run_latency() { timeout -k 10 400 python tools/bench_latency.py --out "$O/latency.json" > "$O/latency.log" 2>&1; }
run_smoke() { timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1; }
run_bench() { timeout -k 10 400 python bench.py > "$O/bench.log" 2>&1; }
run_stats() { timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > "$O/stats.log" 2>&1; }
run_pmc() { local d=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$O/$d" -- python3 bench.py --steps-only --steps 1 --warmup 1 > "$O/$d.log" 2>&1; }
run_sync() { timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/sync" -- python3 tools/bench_sync.py > "$O/sync.log" 2>&1; }
run_ladder() { timeout -k 10 400 python tools/run_ladder_sweep.py --trials 65536 --out "$O/ladder.jsonl" > "$O/ladder.log" 2>&1; }
run_grid() { timeout -k 10 300 python tools/run_acquisition_grid.py --preambles 20000 > "$O/acq_grid.log" 2>&1; }
run_c2() { timeout -k 10 200 python tools/bench_demod.py --mod DQPSK --rate R1_2 --batch 10000 --channel 0 --reps 20 > "$O/c2_demod.log" 2>&1; }
step pytest run_pytest; tail -2 "$O/pytest_gpu.log"
step smoke run_smoke
step bench run_bench; tail -1 "$O/bench.log" | cut -c1-160
step stats run_stats
export RIA_NO_SPLIT=1   # one launch = the whole step (the unit of bench.py's roofline figures)
step fetch run_pmc fetch FETCH_SIZE
step write run_pmc write WRITE_SIZE
unset RIA_NO_SPLIT
step sq run_pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
step sync run_sync
step latency run_latency
step c2 run_c2
step grid run_grid
step ladder run_ladder
ls "$O"
