# the same build, five independent tunings (fresh cache each): how much does the tuner's choice move the result?
set -e
mkdir -p gpurun_out/ts
W=${1:-yolox_s_glfusion_1344x800_bs8}
for i in 1 2 3 4 5; do
  export GLSDET_TUNE_CACHE=/tmp/tc_ts_$i.json; rm -f $GLSDET_TUNE_CACHE
  timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload $W --steps 50 --warmup 10 > gpurun_out/ts/r$i.log 2>&1
  tail -1 gpurun_out/ts/r$i.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('run $i', d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['launches_per_step'])"
done
