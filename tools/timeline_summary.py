#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run of bench.py (three plans in flight): over the LAST `frac` of the trace (the timed
region), how many kernels run at once, how long nothing runs, what runs beside the big MFMA kernels.
usage: timeline_summary.py <dir-with-*_kernel_trace.csv> [window_ms=300] [big_us=80]"""
import csv, glob, sys, collections

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
big_us = float(sys.argv[3]) if len(sys.argv) > 3 else 80.0
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
ev.sort()
# the timed region = the densest window of `frac` milliseconds (graph replays; the eager op-table passes around it are sparser)
W = int(frac * 1e6)
best, bi, j = -1, 0, 0
for i in range(len(ev)):
    while ev[j][0] < ev[i][0] - W:
        j += 1
    if i - j > best:
        best, bi = i - j, j
lo = ev[bi][0]
ev = [e for e in ev if lo <= e[0] <= lo + W]
span = (max(e[1] for e in ev) - ev[0][0]) / 1e3
pts = []
for s, e, n in ev:
    big = (e - s) / 1e3 >= big_us
    pts.append((s, 1, big))
    pts.append((e, -1, big))
pts.sort()
conc = collections.Counter()
cur = curbig = 0
last = pts[0][0]
with_big = collections.Counter()
for t, dlt, big in pts:
    dt = (t - last) / 1e3
    conc[cur] += dt
    if curbig:
        with_big[cur - curbig] += dt
    cur += dlt
    if big:
        curbig += dlt
    last = t
tot_k = sum((e - s) for s, e, n in ev) / 1e3
print("window %.1f ms, %d kernels, sum of kernel time %.1f ms (%.2f x the window)" % (span / 1e3, len(ev), tot_k / 1e3, tot_k / span))
for k in sorted(conc):
    print("  %d kernels running: %5.1f %% of the window" % (k, 100 * conc[k] / span))
tb = sum(with_big.values())
print("a kernel of >= %.0f us is running %.1f %% of the window; beside it run" % (big_us, 100 * tb / span))
for k in sorted(with_big):
    print("  %d other kernels: %5.1f %% of that time" % (k, 100 * with_big[k] / max(tb, 1e-9)))
