set -e
mkdir -p gpurun_out/mp
timeout -k 10 500 python -m pytest tests/test_hip_fuzz.py tests/test_hip_ops.py -m gpu -q -x 2>&1 | tail -3
export GLSDET_TUNE_CACHE=/tmp/tc_mp.json
timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload mp_det_res50_1344x800_bs8 --steps 30 --warmup 8 --op-table gpurun_out/mp/ops.tsv > gpurun_out/mp/bench.log 2>&1
tail -1 gpurun_out/mp/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['all_ops_ms_per_step_eager'], r['launches_per_step'], r['frac'])"
