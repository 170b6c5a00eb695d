#!/usr/bin/env python3
"""Condenses rocprofv3 output (kernel stats + per-dispatch PMC csv) into small per-kernel summaries.
usage: summarize_prof.py <prof_dir> <tag>   -> writes <prof_dir>/<tag>_kernel_stats.csv, <tag>_pmc.json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"pws::conv_mfma_kernel<pws::ConvCfg<([^>]*)>\s*>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        return "conv_mfma_kernel<KS%s,S%s,convT=%s,tile%sx%sx%s,CK%s%s>" % (a[0], a[1], a[3][0], a[6], a[4], a[5], a[7],
                                                                           ",NCHW" if len(a) > 12 and a[12].startswith("t") else "")
    return re.sub(r"\(.*", "", name).replace("pws::", "")[:90]


# ---- kernel stats
stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
rows = []
if stats:
    with open(stats[0]) as f:
        for r in csv.DictReader(f):
            rows.append(r)
    out = os.path.join(d, "%s_kernel_stats.csv" % tag)
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    print("kernel stats ->", out)
    for r in rows[:14]:
        print("%6s calls  avg %10.1f us  %5s%%  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"], short(r["Name"])))
else:
    print("no kernel_stats csv found under", d)

# ---- PMC: aggregate per kernel name: mean counter value per dispatch
pmc = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_grbm"):
    for fn in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for r in csv.DictReader(f):
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {}
for k, cs in pmc.items():
    summ[k] = {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()}
out = os.path.join(d, "%s_pmc.json" % tag)
with open(out, "w") as f:
    json.dump(summ, f, indent=1, sort_keys=True)
print("pmc ->", out)
for k, cs in sorted(summ.items()):
    if "conv_mfma" in k or "grid_sample" in k or "field_head" in k:
        print(k)
        for c, v in sorted(cs.items()):
            print("    %-28s %16.1f  (n=%d)" % (c, v["mean_per_dispatch"], v["dispatches"]))
