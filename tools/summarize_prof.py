#!/usr/bin/env python3
"""Condenses rocprofv3 (rocpd sqlite) output into small per-kernel summaries that can be committed under profiles/.

usage: summarize_prof.py <prof_dir> <tag> [<dest_dir>]
  <prof_dir>/trace/*.db            from  rocprofv3 --kernel-trace --stats
  <prof_dir>/pmc_*/*.db            from  rocprofv3 --pmc ...   (one run per counter group)
writes <dest>/<tag>_kernel_stats.csv and <dest>/<tag>_pmc.json
"""
import csv
import glob
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]
dest = sys.argv[3] if len(sys.argv) > 3 else d
os.makedirs(dest, exist_ok=True)


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"pws::conv_bf16_kernel<pws::BfCfg<([^>]*)>\s*>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        sub = {"0": "", "1": ",convT4", "2": ",dgrad-subpix"}.get(a[3], "," + a[3])
        return "conv_bf16_kernel<k%ss%s%s,tile %sx%sx%s,CK%s>" % (a[0], a[1], sub, a[6], a[4], a[5], a[7])
    m = re.match(r"pws::conv_mfma_kernel<pws::ConvCfg<([^>]*)>\s*>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        sub = {"0": "", "1": ",convT4", "2": ",dgrad-subpix", "true": ",convT4", "false": ""}.get(a[3], "," + a[3])
        return "conv_mfma_kernel<k%ss%s%s,tile %sx%sx%s,CK%s%s>" % (
            a[0], a[1], sub, a[6], a[4], a[5], a[7], ",NCHW" if len(a) > 12 and a[12].startswith("t") else "")
    m = re.match(r"pws::conv_ring_kernel<pws::RgCfg<([^>]*)>,\s*(true|false)(?:,\s*(true|false))?\s*>", name)
    if m:   # persistent LDS-ring bf16 conv (conv_ring.hip): mode, tile, ring depth, epilogue kind (+ sign-bit variant of the data gradient)
        a = [x.strip() for x in m.group(1).split(",")]
        mode = {"0": "k3s1", "1": "convT4", "2": "dgrad-subpix k3s2", "3": "k3s2 planes", "4": "dgrad k4s2 planes", "5": "k5s1 first layer, weights resident"}.get(a[0], a[0])
        return "conv_ring_kernel<%s,tile %sx%sx%s,R%s,%s%s>" % (mode, a[3], a[1], a[2], a[4], "dgrad" if m.group(2) == "true" else "fwd",
                                                                 ",sign bits" if m.group(3) == "true" else "")
    m = re.match(r"(?:pws::)?wino_ring_kernel<(\d+),\s*(\d+),\s*(\d+)>", name)
    if m:   # persistent LDS-ring Winograd kernel (conv_wring.hip): mode, map geometry, ablation mask (0 in the product)
        return "wino_ring_kernel<%s%s%s>" % ({"0": "F(2x2,3x3)", "1": "convT4,F(2x2,2x2),2 classes per unit", "2": "convT4,F(2x2,2x2),1 class per unit"}[m.group(1)],
                                           ",16-wide maps (2 samples per unit)" if m.group(2) == "1" else "", "" if m.group(3) == "0" else ",ablation %s" % m.group(3))
    m = re.match(r"pws::conv_ringf_kernel<pws::RfCfg<([^>]*)>\s*>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        mode = {"0": "k3s1", "1": "convT4", "2": "k3s2 planes"}.get(a[0], a[0])
        return "conv_ringf_kernel<%s,tile %sx32,R%s>" % (mode, a[1], a[2])
    return re.sub(r"\(.*", "", name).replace("pws::", "")[:100]


def db(sub):
    f = glob.glob(os.path.join(d, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


def kernel_stats(sub, suffix):
    c = db(sub)
    if not c:
        return
    agg = defaultdict(list)
    for name, dur in c.execute("select name, duration from kernels"):
        agg[short(name)].append(dur)
    tot = sum(sum(v) for v in agg.values())
    out = os.path.join(dest, "%s_kernel_stats%s.csv" % (tag, suffix))
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot, 3), min(v), max(v)])
    print("kernel stats ->", out)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:16]:
        print("%6d calls  avg %10.1f us  %6.2f%%  %s" % (len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot, k))


kernel_stats("trace", "")
kernel_stats("trace_bf16", "_bf16")
kernel_stats("trace_train", "_train_bf16")
kernel_stats("trace_configs2", "_configs2_bf16")

summ = defaultdict(dict)
for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
    if not os.path.isdir(sub):
        continue
    c = db(os.path.basename(sub))
    if not c:
        continue
    per = defaultdict(lambda: defaultdict(float))  # (kernel, dispatch) -> counter -> summed over instances
    durs = {}
    for name, disp, cn, val, dur in c.execute(
            "select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
        per[(short(name), disp)][cn] += val
        durs[(short(name), disp)] = dur
    bykernel = defaultdict(lambda: defaultdict(list))
    for (k, disp), cs in per.items():
        for cn, val in cs.items():
            bykernel[k][cn].append(val)
        bykernel[k]["_duration_ns_under_pmc"].append(durs[(k, disp)])
    for k, cs in bykernel.items():
        for cn, vals in cs.items():
            summ[k][cn] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
out = os.path.join(dest, "%s_pmc.json" % tag)
try:   # the kernels these counters describe: bench.py quotes counter-derived figures only when this matches its own build
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pwstablenet_amd.build import source_hash
    summ["_meta"] = {"source_hash": source_hash(), "tag": tag}
except Exception as e:
    summ["_meta"] = {"source_hash": None, "error": str(e)}
with open(out, "w") as f:
    json.dump(summ, f, indent=1, sort_keys=True)
print("pmc ->", out)
for k, cs in sorted(summ.items()):
    if k == "_meta":
        continue
    if "conv_" in k or "grid_sample" in k or "field_head" in k or "wino" in k or "wgrad" in k:
        print(k)
        for cn, v in sorted(cs.items()):
            print("    %-28s %18.1f  (n=%d)" % (cn, v["mean_per_dispatch"], v["dispatches"]))
