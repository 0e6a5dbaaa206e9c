set -e
mkdir -p gpurun_out/bn
export GLSDET_TUNE_CACHE=/tmp/tc_a.json
GLSDET_NO_BNECK_FUSION=1 timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 50 --warmup 10 --op-table gpurun_out/bn/ops_off.tsv > gpurun_out/bn/off.log 2>&1
export GLSDET_TUNE_CACHE=/tmp/tc_b.json
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 50 --warmup 10 --op-table gpurun_out/bn/ops_on.tsv > gpurun_out/bn/on.log 2>&1
export GLSDET_TUNE_CACHE=/tmp/tc_c.json
GLSDET_FORCE_BNECK=1 timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 50 --warmup 10 --op-table gpurun_out/bn/ops_force.tsv > gpurun_out/bn/force.log 2>&1
export GLSDET_TUNE_CACHE=/tmp/tc_a.json
GLSDET_NO_BNECK_FUSION=1 timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/bn/off2.log 2>&1
export GLSDET_TUNE_CACHE=/tmp/tc_b.json
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/bn/on2.log 2>&1
for f in off on force off2 on2; do tail -1 gpurun_out/bn/$f.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['all_ops_ms_per_step_eager'], r['launches_per_step'], r['frac'])"; done
