#!/usr/bin/env python3
"""Timeline of ONE steady-state generator forward out of a rocprofv3 --kernel-trace database (rocpd sqlite): start relative to the first
kernel of the forward, duration, queue, kernel -- and how much of the wall time of the forward no kernel runs / one queue runs / both run.
usage: timeline_prof.py <dir with the .db> [marker substring of the forward's first kernel] [which forward from the end, default 3]"""
import glob
import os
import sqlite3
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "first_kernel"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
c = sqlite3.connect(f[0])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("# columns of `kernels`:", cols)
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else ("stream_id" if "stream_id" in cols else None))
rows = list(c.execute("select name, start, end%s from kernels order by start" % (", " + qcol if qcol else "")))
starts = [i for i, r in enumerate(rows) if marker in r[0]]
if len(starts) < back + 1:
    sys.exit("marker %r found %d times" % (marker, len(starts)))
i0, i1 = starts[-back - 1], starts[-back]
sel = rows[i0:i1]
t0 = sel[0][1]
ev = []
for r in sel:
    name = r[0].replace("pws::", "")
    name = name if len(name) < 90 else name[:87] + "..."
    print("%9.1f us  %8.1f us  q %-6s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", name))
    ev += [(r[1], 1), (r[2], -1)]
ev.sort()
depth, last, hist = 0, ev[0][0], {}
for t, dlt in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth, last = depth + dlt, t
wall = rows[i1][1] - t0
print("# forward: %d kernels, %.1f us from its first kernel to the next forward's; sum of durations %.1f us" % (len(sel), wall / 1e3, sum(r[2] - r[1] for r in sel) / 1e3))
print("# time with k kernels running:", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())}, "(+ %.1f us idle before the next forward)" % ((rows[i1][1] - max(r[2] for r in sel)) / 1e3))
