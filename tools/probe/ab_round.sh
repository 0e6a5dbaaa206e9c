set -e
mkdir -p gpurun_out/ab
timeout -k 10 900 python -m pytest tests/test_hip_fuzz.py tests/test_hip_ops.py tests/test_hip_model.py -m gpu -q -x 2>&1 | tail -3
export GLSDET_TUNE_CACHE=/tmp/tc_gl.json
timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --steps 50 --warmup 10 --op-table gpurun_out/ab/ops_gl.tsv > gpurun_out/ab/gl.log 2>&1
export GLSDET_TUNE_CACHE=/tmp/tc_mp.json
timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload mp_det_res50_1344x800_bs8 --steps 30 --warmup 8 --op-table gpurun_out/ab/ops_mp.tsv > gpurun_out/ab/mp.log 2>&1
for f in gl mp; do tail -1 gpurun_out/ab/$f.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['all_ops_ms_per_step_eager'], r['launches_per_step'], r['frac'], r['frac_end_to_end'])"; done
