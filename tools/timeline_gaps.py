#!/usr/bin/env python3
"""Idle time of the GPU between the kernels of a rocprofv3 --kernel-trace database: for the window between the LAST two occurrences of a marker
kernel (one training step, one forward ...), the time with 0 / 1 / 2 kernels running and the largest gaps with the kernels on either side.
usage: timeline_gaps.py <dir with the .db> <marker substring> [how many gaps to list, default 25]"""
import glob
import os
import sqlite3
import sys

d, marker = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
f = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
c = sqlite3.connect(f[0])
rows = list(c.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if marker in r[0]]
# the marker may occur several times per step: take the last two occurrences that are at least 5 ms apart
i1 = starts[-1]
i0 = next(i for i in reversed(starts) if rows[i1][1] - rows[i][1] > 5e6)
sel = rows[i0:i1]
t0 = sel[0][1]
ev = []
for r in sel:
    ev += [(r[1], 1), (r[2], -1)]
ev.sort()
depth, last, hist = 0, ev[0][0], {}
for t, dlt in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth, last = depth + dlt, t
print("# window: %d kernels, %.2f ms wall, sum of durations %.2f ms" % (len(sel), (rows[i1][1] - t0) / 1e6, sum(r[2] - r[1] for r in sel) / 1e6))
print("# time with k kernels running (us):", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
gaps, end = [], sel[0][2]
prev = sel[0][0]
for r in sel[1:]:
    if r[1] > end:
        gaps.append((r[1] - end, (end - t0) / 1e3, prev, r[0]))
    if r[2] > end:
        end, prev = r[2], r[0]
gaps.sort(reverse=True)
sh = lambda n: n.replace("pws::", "").replace("void ", "")[:60]
for g, at, a, b in gaps[:top]:
    print("%8.1f us idle at %9.1f us   after %-60s before %s" % (g / 1e3, at, sh(a), sh(b)))
print("# gaps > 20 us: %d, total %.1f us; gaps 5..20 us: %d, total %.1f us; gaps < 5 us: %d, total %.1f us" % (
    sum(1 for g in gaps if g[0] > 2e4), sum(g[0] for g in gaps if g[0] > 2e4) / 1e3,
    sum(1 for g in gaps if 5e3 <= g[0] <= 2e4), sum(g[0] for g in gaps if 5e3 <= g[0] <= 2e4) / 1e3,
    sum(1 for g in gaps if g[0] < 5e3), sum(g[0] for g in gaps if g[0] < 5e3) / 1e3))
