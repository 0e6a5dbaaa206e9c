#!/usr/bin/env python3
"""Print the kernel sequence around __amd_rocclr_copyBuffer launches from a rocprofv3 kernel trace."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"][:70] for r in rows]
idx = [i for i, n in enumerate(names) if "copyBuffer" in n]
print("kernels", len(rows), "copyBuffer", len(idx))
# show the LAST 3 occurrences' neighbourhoods and a histogram of what precedes/follows
import collections
prev = collections.Counter(names[i - 1] for i in idx if i > 0)
nxt = collections.Counter(names[i + 1] for i in idx if i + 1 < len(names))
print("preceded by:", prev.most_common(8))
print("followed by:", nxt.most_common(8))
last = idx[-1]
for i in range(max(0, last - 130), min(len(rows), last + 5)):
    r = rows[i]
    print(i, r.get("Queue_Id", ""), r.get("Stream_Id", ""), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), names[i])
