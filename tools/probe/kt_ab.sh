set -u
export TMPDIR=/tmp
out=gpurun_out/r03_v; mkdir -p $out
export GLSDET_TUNE_CACHE=$PWD/$out/tune_cache.json
python bench.py --no-secondary --steps 20 --warmup 5 --windows 1 --no-cpu-baseline > $out/warm.log 2>&1
for m in m16 m32; do
  if [ $m = m32 ]; then export GLSDET_NO_M16=1; export GLSDET_TUNE_CACHE=$PWD/$out/tune_cache32.json; fi
  rm -rf $out/kt_$m
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$m -- python bench.py --no-secondary --steps 40 --warmup 5 --no-cpu-baseline --settle 0 --windows 1 > $out/kt_$m.log 2>&1
  python tools/prof_summary.py $out/kt_$m 58 > $out/summary_$m.txt
  rm -rf $out/kt_$m
  grep "^{\"metric\"" $out/kt_$m.log | tail -1 | cut -c1-200
done
