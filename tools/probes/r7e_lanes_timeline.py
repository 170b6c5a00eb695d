#!/usr/bin/env python3
"""kernel timeline of the steady state of bench.py --in-flight 2 out of a rocprofv3 --kernel-trace db: a 9 ms window, per queue: start, duration, kernel (short),
and the time with 0 / 1 / 2+ kernels running"""
import glob, os, re, sqlite3, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)
c = sqlite3.connect(f[0])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(c.execute("select name, start, end, %s from kernels order by start" % qcol))
firsts = [r[1] for r in rows if "wino5_first" in r[0]]
lo = firsts[40]   # the 41st forward of the run (steady state of the timed loop); window [lo, lo + 9 ms]
sel = [r for r in rows if lo <= r[1] < lo + 9e6]
qs = sorted(set(r[3] for r in sel))
def short(n):
    n = re.sub(r"^void ", "", n).replace("pws::", "")
    n = re.sub(r"<.*", "", n)
    return n[:28]
for r in sel:
    print("%9.1f us %8.1f us  q%d %s%s" % ((r[1] - lo) / 1e3, (r[2] - r[1]) / 1e3, qs.index(r[3]), "    " * qs.index(r[3]), short(r[0])))
ev = sorted([(r[1], 1) for r in sel] + [(r[2], -1) for r in sel])
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last); depth += d; last = t
print("# time with k kernels running (us):", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
