#!/bin/bash
# kernel timeline of the default bench workload with three plans in flight (GPU box)
set -u
export TMPDIR=/tmp
out=gpurun_out/${1:-timeline}; mkdir -p $out
export GLSDET_TUNE_CACHE=$PWD/$out/tune_cache.json
python bench.py --no-secondary --steps 20 --warmup 5 --windows 1 --no-cpu-baseline > $out/warm.log 2>&1
rm -rf $out/kt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python bench.py --no-secondary --steps 200 --warmup 20 --no-cpu-baseline --settle 0 --windows 1 ${2:-} > $out/kt.log 2>&1
python tools/timeline_summary.py $out/kt 300 80 > $out/timeline.txt
grep "^{\"metric\"" $out/kt.log | tail -1 | cut -c1-220
cat $out/timeline.txt
rm -rf $out/kt
