#!/usr/bin/env python3
"""Developer aid: turn the rocprofv3 outputs of one measurement session (gpurun_out/) into the committed
profiles/<tag>_* files.  usage: tools/make_profile_summary.py <tag> <stats_dir> <fetch_dir> <write_dir> <sq_dir> <bench_json>"""
import collections, csv, glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from ria_amd.srchash import csrc_sha256  # noqa: E402
# the PMC passes run `bench.py --steps-only --steps 1 --warmup 1`: two whole steps and nothing else on the GPU
BATCH = int(os.environ.get("RIA_PROFILE_BATCH", "100000"))   # bench.py's default frames per step
META = {"tag": None, "source_sha256": csrc_sha256(), "steps_counted": 2, "frames_per_step": BATCH, "frames_per_launch": BATCH,
        "note": "source_sha256 = sha256 over ria_amd/csrc/* at the time of the measurement (ria_amd/srchash.py); bench.py quotes "
                "these figures only while the sources are unchanged"}
tag, stats_dir, fetch_dir, write_dir, sq_dir, bench_json = sys.argv[1:7]
META["tag"] = tag
P = os.path.join(root, "profiles")


def one(pattern):
    return sorted(glob.glob(pattern))[-1]


def short(k):
    return k.split("(")[0].replace("void ", "")


def agg(path, counters):
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set(); dur = collections.defaultdict(float)
    for r in rows:
        k = short(r["Kernel_Name"])
        if r["Counter_Name"] in counters:
            a[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1; dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return a, n, dur


shutil.copy(one(os.path.join(stats_dir, "*", "*kernel_stats.csv")), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
json.dump(bench, open(os.path.join(P, f"{tag}_bench.json"), "w"))
f, nf, fdur = agg(one(os.path.join(fetch_dir, "*", "*counter_collection.csv")), {"FETCH_SIZE"})
w, nw, _ = agg(one(os.path.join(write_dir, "*", "*counter_collection.csv")), {"WRITE_SIZE"})
traffic = {}
for k in f:
    if "ria::" in k:
        fe = f[k]["FETCH_SIZE"] / nf[k]; wr = w.get(k, {}).get("WRITE_SIZE", 0.0) / max(1, nw.get(k, 1))
        # rocprofv3 reports KB; gfx950: FETCH_SIZE tallies 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM section)
        # launches_per_step: the PMC passes run two whole steps (warm-up + timed); avg_launch_ms: this kernel's mean duration in the FETCH pass
        traffic[k] = {"launches": nf[k], "launches_per_step": nf[k] / META["steps_counted"], "avg_launch_ms": round(fdur[k] / nf[k] * 1e-6, 4),
                      "fetch_kb_raw": round(fe, 1), "write_kb_raw": round(wr, 1), "hbm_bytes_per_launch": round(fe * 1024 * 2 + wr * 1024)}
traffic["_meta"] = META
json.dump(traffic, open(os.path.join(P, f"{tag}_hbm_traffic_pmc.json"), "w"), indent=1)
del traffic["_meta"]
a, n, dur = agg(one(os.path.join(sq_dir, "*", "*counter_collection.csv")), {"SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_LDS"})
sq = {}
for k, v in a.items():
    if "ria::" not in k:
        continue
    d = dur[k] / n[k] * 1e-9; cyc = d * 2.4e9
    # valu_active_quadcycle_frac: per-wave SQ_ACTIVE_INST_VALU quad-cycles summed over the resident waves / SIMD cycles (it counts a wave
    # as active while its instruction is in flight, so with several waves per SIMD it overstates the pipe's occupancy);
    # valu_issue_frac_2cyc: SQ_INSTS_VALU x 2 cycles (wave64 on SIMD-32) / SIMD cycles - the issue-slot share bench.py quotes
    sq[k] = {"launches": n[k], "avg_ms": round(d * 1e3, 3), "valu_active_quadcycle_frac": round(v["SQ_ACTIVE_INST_VALU"] / n[k] * 4 / (1024 * cyc), 3),
             "valu_issue_frac_2cyc": round(v["SQ_INSTS_VALU"] / n[k] * 2 / (1024 * cyc), 3),
             "lds_idx_active": v["SQ_LDS_IDX_ACTIVE"] / n[k], "lds_bank_conflict": v["SQ_LDS_BANK_CONFLICT"] / n[k],
             "lds_busy_frac": round(v["SQ_LDS_IDX_ACTIVE"] / n[k] / (256 * cyc), 3),
             "lds_bank_conflict_share": round(v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"]), 3),
             "valu_insts": v["SQ_INSTS_VALU"] / n[k], "lds_insts": v["SQ_INSTS_LDS"] / n[k]}
sq["_meta"] = META
json.dump(sq, open(os.path.join(P, f"{tag}_sq_utilisation_pmc.json"), "w"), indent=1)
del sq["_meta"]
rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"))))
L = [f"# {tag} — bench, kernel trace and PMC summaries\n",
     "Commands (MI355X box, repo root, after `cd /tmp && export TMPDIR=/tmp`):\n",
     f"* `python bench.py` -> `{tag}_bench.json`",
     f"* `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline` -> `{tag}_bench_kernel_stats.csv`",
     f"* `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --steps-only --steps 1 --warmup 1`, same with `WRITE_SIZE` (separate passes) -> `{tag}_hbm_traffic_pmc.json` (both passes with `RIA_NO_SPLIT=1`, so one launch = one whole {BATCH}-frame step, the unit `bench.py`'s roofline uses; FETCH_SIZE x 2 for gfx950; Infinity-Cache hits are counted)",
     f"* `rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 bench.py ...` -> `{tag}_sq_utilisation_pmc.json`\n",
     f"Bench line: **{bench['value']:.0f} frames/s** on 1 MI355X ({bench['ms_per_step']} ms per {BATCH}-frame step); reference CPU path on the same box: {bench['cpu_baseline']['value']} frames/s on {bench['cpu_baseline']['cores']} threads ({bench['cpu_baseline']['kind']}).\n",
     "| kernel | calls | avg ms | % of GPU time |", "|---|---|---|---|"]
for r in rows[:16]:
    L.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |")
L += ["\nVALU / LDS utilisation (PMC; concurrent kernels of the two streams share the machine, so per-kernel fractions are of the whole chip over that kernel's own duration):\n",
      "| kernel | avg ms | VALU issue (2-cycle model) | LDS busy | LDS bank-conflict share |", "|---|---|---|---|---|"]
for k, v in sorted(sq.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"])[:8]:
    L.append(f"| `{k}` | {v['avg_ms']} | {v['valu_issue_frac_2cyc']} | {v['lds_busy_frac']} | {v['lds_bank_conflict_share']} |")
L += ["\nHBM-side traffic per launch (PMC, corrected):\n", "| kernel | launches | MB per launch |", "|---|---|---|"]
for k, v in traffic.items():
    L.append(f"| `{k}` | {v['launches']} | {v['hbm_bytes_per_launch'] / 1e6:.1f} |")
open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(L) + "\n")
print("wrote profiles/%s_*" % tag)
