#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace csv (…_kernel_trace.csv) of tools/configs2_step.py and reports, for the LAST training step, how the
wall time splits: nothing running / only "short" kernels (< 40 us) running / at least one long kernel running; plus the ten largest gaps.
usage: python tools/timeline_gaps.py <kernel_trace.csv> [steps_in_trace]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
# the last step: from the last 'u8_normalize' pair backwards
starts = [i for i, e in enumerate(ev) if "u8_normalize" in e[2]]
# 4 normalize launches per step
first = starts[-4] if len(starts) >= 4 else 0
step = ev[first:]
t0, t1 = step[0][0], max(e[1] for e in step)
print("last step: %d launches, %.2f ms wall, %.2f ms summed kernel time" % (len(step), (t1 - t0) / 1e6, sum(e[1] - e[0] for e in step) / 1e6))
pts = []
for s, e, n, q in step:
    long_ = (e - s) >= 40000
    pts.append((s, 1, long_)), pts.append((e, -1, long_))
pts.sort()
run_long = run_short = 0
last = t0
acc = {"idle": 0, "short only": 0, "long": 0}
gaps = []
for t, d, lg in pts:
    dur = t - last
    if dur > 0:
        key = "long" if run_long > 0 else ("short only" if run_short > 0 else "idle")
        acc[key] += dur
        if key == "idle":
            gaps.append((dur, last - t0))
    if lg:
        run_long += d
    else:
        run_short += d
    last = t
for k, v in acc.items():
    print("  %-10s %7.2f ms" % (k, v / 1e6))
print("largest idle gaps (us @ ms into the step):", ", ".join("%.0f@%.1f" % (g / 1e3, at / 1e6) for g, at in sorted(gaps, reverse=True)[:12]))
qs = {}
for s, e, n, q in step:
    qs.setdefault(q, [0, 0])
    qs[q][0] += 1
    qs[q][1] += e - s
print("per queue:", {q: "%d launches, %.1f ms" % (v[0], v[1] / 1e6) for q, v in qs.items()})
