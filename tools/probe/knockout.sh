#!/bin/bash
# What each class of ops is worth in the CONCURRENT regime (three batches in flight): the default bench with that class
# recorded as no-ops (GLSDET_SKIP_OPS; results are garbage, only the wall clock counts).  usage: knockout.sh OUTDIR
out=${1:-gpurun_out/knock}
mkdir -p $out
export GLSDET_TUNE_CACHE=$out/tune.json
run() {
  tag=$1; shift
  GLSDET_SKIP_OPS="$1" python bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 --max-det 30000 > $out/$tag.json 2> $out/$tag.err
  python - "$tag" "$out/$tag.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-14s %8.1f img/s  %.4f ms/step  (eager sum of all ops %.3f ms)" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["all_ops_ms_per_step_eager"]), flush=True)
except Exception as e:
    print("%-14s FAILED %s" % (sys.argv[1], e), flush=True)
PY
}
run base ""
run no_7x7 "7x7"
run no_5x5 "5x5"
run no_bighalo "7x7,5x5,cin128 cout256,ring_k64_multi[2],ring8"
run no_stem "focus_stem"
run no_front "focus_stem,cin32 cout64,cin64 cout64,cin32 cout32,cin64 cout128"
run no_1x1 " 1x1 s1"
run no_s2 "3x3 s2"
run no_nms "nms"
run no_bneck "conv_bneck"
run no_nonlocal "nonlocal"
run base2 ""
