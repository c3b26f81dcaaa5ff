#!/usr/bin/env python3
"""Developer aid: per-kernel sums of the counters in a rocprofv3 --pmc counter_collection.csv (per launch averages).
usage: tools/pmc_table.py <counter_collection.csv> [kernel-name-substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "ria::"
a = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set(); dur = collections.defaultdict(float)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ria::", "")
    if pat.replace("ria::", "") not in k and pat != "ria::":
        continue
    a[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"])); n[k] += 1; dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k in sorted(a, key=lambda k: -dur[k]):
    print(f"{k[:50]:50s} launches {n[k]:4d} avg_ms {dur[k] / n[k] * 1e-6:8.3f} " + " ".join(f"{c}={v / n[k]:.4g}" for c, v in sorted(a[k].items())))
