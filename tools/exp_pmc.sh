# Developer aid: SQ counters of the core microbenchmark (tools/bench_core.py) for one variant in build/ab
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$1; O=gpurun_out/pmc_$V; rm -rf $O; mkdir -p $O
export RIA_GPU_LIB=$GRAFT_REPO_ROOT/build/ab/$V.so
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/a -- python3 tools/bench_core.py R1_2 65536 > $O/a.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAVES --output-format csv -d $O/b -- python3 tools/bench_core.py R1_2 65536 > $O/b.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    acc = collections.defaultdict(float); n = 0
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "fast_rows" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    nl = max(1, sum(1 for f in glob.glob("$O/%s/**/*kernel_trace.csv" % sub, recursive=True) for r in csv.DictReader(open(f)) if "fast_rows" in r["Kernel_Name"]))
    for k, v in sorted(acc.items()): print("$V", sub, k, "%.4g per launch" % (v / nl))
PY
