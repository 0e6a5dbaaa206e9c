#!/bin/bash
# Run ON THE GPU BOX (via gpurun): only the two HBM-traffic PMC passes of tools/profile_round.sh + the per-kernel table.
# usage: tools/pmc_traffic.sh <tag> [workload]     outputs under gpurun_out/<tag>/
set -u
tag=${1:-pmc}
WL=${2:-yolox_s_glfusion_1344x800_bs8}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
export GLSDET_TUNE_CACHE=$PWD/$out/tune_cache.json
STEPS=10; WARM=3
timeout -k 10 300 python bench.py --no-secondary --workload $WL --steps 20 --warmup 5 --windows 1 --no-cpu-baseline --op-table $out/ops.tsv > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
rm -rf $out/pmc_f; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python bench.py --no-secondary --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --settle 0 --windows 1 --streams 1 --no-graph > $out/pmc_f.log 2>&1
rm -rf $out/pmc_w; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -- python bench.py --no-secondary --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --settle 0 --windows 1 --streams 1 --no-graph > $out/pmc_w.log 2>&1
python tools/traffic_summary.py $out $WL > $out/traffic.json
python tools/traffic_by_kernel.py $out 2.0 > $out/traffic_by_kernel.txt
rm -rf $out/pmc_f $out/pmc_w
head -60 $out/traffic_by_kernel.txt
