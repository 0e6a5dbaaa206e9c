# A/B on ONE box: bench variants back to back, 2 rounds.   usage: tools/ab.sh <workload>
wl=${1:-yolox_s_glfusion_1344x800_bs8}
mkdir -p gpurun_out/ab
for round in 1 2; do
for v in base nogroup nout; do
  case $v in base) E="";; nogroup) E="GLSDET_NO_GROUP=1";; nout) E="GLSDET_NO_UT=1";; esac
  env $E timeout -k 10 300 python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/ab/$v.$round.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/ab/$v.$round.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$wl $v round $round:", d.get("value"), d.get("ms_per_step"), d.get("roofline",{}).get("conv_ms_per_step"), d.get("roofline",{}).get("all_ops_ms_per_step_eager"))
PY
done; done
