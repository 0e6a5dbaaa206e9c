#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench + rocprofv3 kernel stats + HBM traffic counters.
# usage: tools/profile_round.sh <tag> [workload]     outputs under gpurun_out/<tag>/
set -u
tag=${1:-r01}
WL=${2:-yolox_s_glfusion_1344x800_bs8}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
export GLSDET_TUNE_CACHE=$PWD/$out/tune_cache.json
rm -f $GLSDET_TUNE_CACHE
STEPS=20; WARM=5
timeout -k 10 400 python bench.py --no-secondary --workload $WL --steps 50 --warmup 10 --op-table $out/ops.tsv > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
tail -1 $out/bench.log > $out/bench.json
rm -rf $out/kt; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python bench.py --no-secondary --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --settle 0 --windows 1 > $out/kt.log 2>&1
python tools/prof_summary.py $out/kt $((STEPS+WARM+13)) > $out/kernel_stats_summary.txt
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
# the same kernels on ONE stream without graph replay: these averages are the ones bench.py's roofline
# (HIP events around every op of an eager replay) must agree with -- with three batches in flight the
# concurrent kernels stretch each other
rm -rf $out/kt1; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt1 -- python bench.py --no-secondary --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --settle 0 --windows 1 --streams 1 --no-graph > $out/kt1.log 2>&1
python tools/prof_summary.py $out/kt1 $((STEPS+WARM+13)) > $out/kernel_stats_single_stream_summary.txt
cp $(find $out/kt1 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_single_stream.csv
grep "^{\"metric\"" $out/kt1.log | tail -1 > $out/bench_single_stream.json
rm -rf $out/pmc_f; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python bench.py --no-secondary --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --settle 0 --windows 1 > $out/pmc_f.log 2>&1
rm -rf $out/pmc_w; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -- python bench.py --no-secondary --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --settle 0 --windows 1 > $out/pmc_w.log 2>&1
python tools/traffic_summary.py $out $WL > $out/traffic.json
cat $out/traffic.json
python tools/traffic_by_kernel.py $out 2.0 > $out/traffic_by_kernel.txt
rm -rf $out/kt $out/kt1 $out/pmc_f $out/pmc_w
head -16 $out/kernel_stats_summary.txt
