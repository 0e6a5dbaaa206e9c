#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: per-step time per kernel.
usage: prof_summary.py <dir-with-*_kernel_stats.csv> <steps-profiled> [filter]"""
import csv, glob, sys, re
d, steps = sys.argv[1], float(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
f = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms/step %.3f" % (tot / 1e6 / steps))
for r in rows:
    name = r["Name"]
    m = re.match(r"_ZN6glsdet\d+(\w+?)I(.*)EEv", name)
    short = name if not m else m.group(1) + "<" + m.group(2) + ">"
    short = short.replace("DF16_", "h,").replace("Li", "").replace("ELi", ",").replace("E", ",")
    if flt and flt not in name:
        continue
    print("%8.4f ms/step  calls/step %6.2f  avg %8.2f us  %s" % (
        float(r["TotalDurationNs"]) / 1e6 / steps, float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, short[:110]))
