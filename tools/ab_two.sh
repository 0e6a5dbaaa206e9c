#!/bin/bash
# A/B on ONE box: the working-tree library against glsdet_amd/lib/ab/libglsdet_hip.so (tools/ab_lib.sh), alternating.
#   usage: tools/ab_two.sh <rounds> <workload> [workload ...]
rounds=$1; shift
mkdir -p gpurun_out/ab
for wl in "$@"; do
for round in $(seq 1 $rounds); do
for v in new old; do
  case $v in new) E="GLSDET_X=0";; old) E="GLSDET_LIB_PATH=$PWD/glsdet_amd/lib/ab/libglsdet_hip.so";; esac
  env $E timeout -k 10 300 python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/ab/$v.$round.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/ab/$v.$round.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$wl $v round $round:", d.get("value"), d.get("ms_per_step"), "conv eager ms", d.get("roofline",{}).get("conv_ms_per_step"), "all eager", d.get("roofline",{}).get("all_ops_ms_per_step_eager"))
PY
done; done; done
