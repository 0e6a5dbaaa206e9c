# batch size x batches in flight: would a step's 8 images run faster as two half batches on more streams?
set -e
mkdir -p gpurun_out/sb
for cfg in "8 3" "4 3" "4 4" "4 6" "4 8" "16 2" "16 3" "8 3"; do
  set -- $cfg
  export GLSDET_TUNE_CACHE=/tmp/tc_sb_$1.json
  timeout -k 10 400 python bench.py --no-secondary --no-cpu-baseline --workload yolox_s_glfusion_1344x800_bs$1 --steps 60 --warmup 10 --streams $2 > gpurun_out/sb/b$1_s$2.log 2>&1
  tail -1 gpurun_out/sb/b$1_s$2.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bs $1 streams $2:', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done
