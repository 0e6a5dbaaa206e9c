# A/B of glsdet_amd/lib/ab/libglsdet_hip.so (A = built by tools/ab_lib.sh) against the working-tree library (B), one call
set -e
W=${1:-yolox_s_glfusion_1344x800_bs8}
mkdir -p gpurun_out/abl
for rep in 1 2; do
  export GLSDET_TUNE_CACHE=/tmp/tc_A.json
  GLSDET_LIB_PATH=$PWD/glsdet_amd/lib/ab/libglsdet_hip.so timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps 50 --warmup 10 --op-table gpurun_out/abl/ops_A.tsv > gpurun_out/abl/A$rep.log 2>&1
  export GLSDET_TUNE_CACHE=/tmp/tc_B.json
  timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps 50 --warmup 10 --op-table gpurun_out/abl/ops_B.tsv > gpurun_out/abl/B$rep.log 2>&1
done
for f in A1 B1 A2 B2; do tail -1 gpurun_out/abl/$f.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['all_ops_ms_per_step_eager'], r['launches_per_step'], r['frac'], r['frac_end_to_end'])"; done
