set -e
mkdir -p gpurun_out/gn
W=mp_det_res50_1344x800_bs8
export GLSDET_TUNE_CACHE=/tmp/tc_a.json
timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps 30 --warmup 8 --op-table gpurun_out/gn/ops_off.tsv > gpurun_out/gn/off.log 2>&1
export GLSDET_TUNE_CACHE=/tmp/tc_b.json
GLSDET_GN_FUSION=1 timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps 30 --warmup 8 --op-table gpurun_out/gn/ops_on.tsv > gpurun_out/gn/on.log 2>&1
export GLSDET_TUNE_CACHE=/tmp/tc_a.json
timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps 30 --warmup 8 > gpurun_out/gn/off2.log 2>&1
export GLSDET_TUNE_CACHE=/tmp/tc_b.json
GLSDET_GN_FUSION=1 timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps 30 --warmup 8 > gpurun_out/gn/on2.log 2>&1
for f in off on off2 on2; do tail -1 gpurun_out/gn/$f.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['all_ops_ms_per_step_eager'], r['launches_per_step'], r['frac'])"; done
grep -i "groupnorm\|gn stats" gpurun_out/gn/ops_on.tsv
