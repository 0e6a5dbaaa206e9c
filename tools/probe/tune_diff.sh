#!/bin/bash
# two fresh tunings of the default workload on one box: which layers does the tuner decide differently, and what does it cost?
out=gpurun_out/${1:-tune_diff}; mkdir -p $out
for i in 1 2 3; do
  export GLSDET_TUNE_CACHE=$PWD/$out/tune$i.json; rm -f $GLSDET_TUNE_CACHE
  timeout -k 10 250 python bench.py --no-secondary --no-cpu-baseline --windows 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tuning $i', d['value'], d['ms_per_step'], d['roofline']['launches_per_step'])"
  # the same table again (no tuning): run-to-run spread of ONE selection
  timeout -k 10 250 python bench.py --no-secondary --no-cpu-baseline --windows 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   replayed', d['value'], d['ms_per_step'])"
done
python - <<PY
import json
t=[json.load(open("$out/tune%d.json"%i))["f16"] for i in (1,2,3)]
keys=set(t[0])|set(t[1])|set(t[2])
n=0
for k in sorted(keys):
    v=[x.get(k) for x in t]
    if len(set(v))>1:
        n+=1
        kk=json.loads(k)
        print(v, kk[:1], kk[1:12])
print(n, "of", len(keys), "keys differ")
PY
