set -e
mkdir -p gpurun_out/st
export GLSDET_TUNE_CACHE=/tmp/tc_st.json
for W in yolox_s_glfusion_1344x800_bs8 mp_det_res50_gl_1344x800_bs8; do
for s in 3 2 4 5 3; do
  timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps 60 --warmup 10 --streams $s > gpurun_out/st/s$s.log 2>&1
  tail -1 gpurun_out/st/s$s.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$W streams $s', d['value'], d['ms_per_step'])"
done; done
