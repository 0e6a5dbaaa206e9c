#!/usr/bin/env python3
"""Per-kernel HBM traffic from the two rocprofv3 PMC passes of tools/profile_round.sh (kept with KEEP_PMC=1):
launches grouped by (kernel name, grid size), mean FETCH_SIZE (x2: the gfx950 correction for wide coalesced reads) and
WRITE_SIZE per launch in MB.  usage: tools/traffic_by_kernel.py gpurun_out/<tag> [min_MB]"""
import collections, csv, glob, re, sys

d = sys.argv[1]
min_mb = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"glsdet::", "", n)
    n = n.replace("_Float16", "h").replace("float", "f")
    return n[:70]


def collect(sub, counter):
    out = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = (short(r["Kernel_Name"]), int(r.get("Grid_Size", 0) or 0))
            out[k][0] += float(r["Counter_Value"]) * 1024.0
            out[k][1] += 1
    return out


F, W = collect("pmc_f", "FETCH_SIZE"), collect("pmc_w", "WRITE_SIZE")
rows = []
for k in set(F) | set(W):
    f, nf = F.get(k, [0.0, 0])
    w, nw = W.get(k, [0.0, 0])
    n = max(nf, nw, 1)
    rows.append((2.0 * f / max(nf, 1) / 1e6, w / max(nw, 1) / 1e6, n, k))
rows.sort(key=lambda r: -(r[0] + r[1]) * r[2])
print("%9s %9s %9s %6s  kernel (grid)" % ("fetch MB", "write MB", "total MB", "calls"))
for f, w, n, k in rows:
    if f + w >= min_mb:
        print("%9.1f %9.1f %9.1f %6d  %s (%d)" % (f, w, f + w, n, k[0], k[1]))
