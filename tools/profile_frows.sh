#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of the kernels either side of the forward pass:
# GPU preprocess, the UFPMP second stage, bbox COCOeval.   usage: bash tools/profile_frows.sh <tag>
set -u
tag=${1:-r01_frows}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for t in pre_bench two_stage_bench eval_bench; do
  rm -rf $out/kt_$t
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$t -- python tools/$t.py > $out/$t.log 2>&1
  f=$(find $out/kt_$t -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    cp $f $out/${t}_kernel_stats.csv
    python - "$f" > $out/${t}_kernel_stats_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "glsdet" in r["Name"] or "kernel" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
    print("%10.1f us total  calls %6s  avg %9.2f us  %s" % (float(r["TotalDurationNs"]) / 1e3, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:120]))
PY
  fi
  rm -rf $out/kt_$t
  grep -v amdgpu.ids $out/$t.log | tail -8
done
