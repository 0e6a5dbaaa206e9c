#!/usr/bin/env python3
"""HBM traffic of the conv kernels from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in KiB).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide
coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
Per-launch mean over all conv launches of the run (autotune is off in these passes: the
tuning cache is loaded), times launches per step = bytes per step."""
import csv, glob, json, sys
d = sys.argv[1]


def per_launch(sub, counter):
    tot, n = 0.0, 0
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (d, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("conv" in r["Kernel_Name"] or "focus_stem" in r["Kernel_Name"]):
                tot += float(r["Counter_Value"]); n += 1
    return tot, n


fk, fn = per_launch("pmc_f", "FETCH_SIZE")
wk, wn = per_launch("pmc_w", "WRITE_SIZE")
res = {"workload": sys.argv[2] if len(sys.argv) > 2 else "yolox_s_glfusion_1344x800_bs8", "conv_launches_counted": fn,
       "fetch_bytes_per_launch_corrected": 2.0 * fk * 1024 / max(fn, 1),
       "write_bytes_per_launch": wk * 1024 / max(wn, 1)}
res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch_corrected"] + res["write_bytes_per_launch"]
print(json.dumps(res))
