# usage: ab_env.sh ENVVAR [workload]   -- A/B of one opt-out switch inside one call (off = ENVVAR=1).  Both sides share ONE
# tune cache (populated by an untimed first run of each side), so that tuner noise -- near-tie variants flip between runs
# and move a workload by up to 1.5 % -- does not decide the comparison.
set -e
V=$1; W=${2:-yolox_s_glfusion_1344x800_bs8}
mkdir -p gpurun_out/abenv
export GLSDET_TUNE_CACHE=/tmp/tc_shared_$$.json
env $V=1 timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps 5 --warmup 2 > /dev/null 2>&1
timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps 5 --warmup 2 > /dev/null 2>&1
for rep in 1 2; do
  env $V=1 timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps ${STEPS:-50} --warmup 10 --op-table gpurun_out/abenv/ops_off.tsv > gpurun_out/abenv/off$rep.log 2>&1
  timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps ${STEPS:-50} --warmup 10 --op-table gpurun_out/abenv/ops_on.tsv > gpurun_out/abenv/on$rep.log 2>&1
done
for f in off1 on1 off2 on2; do tail -1 gpurun_out/abenv/$f.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['all_ops_ms_per_step_eager'], r['launches_per_step'], r['frac'], r['frac_end_to_end'])"; done
head -4 gpurun_out/abenv/ops_on.tsv
