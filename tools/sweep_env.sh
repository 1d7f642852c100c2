#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 v3" label-regex : runs bench.py once per value and prints ms/step + the avg us of matching kernels
VAR=$1; VALS=$2; PAT=$3
for v in $VALS; do
  env $VAR=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --launch-dump gpurun_out/l_$v.jsonl > gpurun_out/b_$v.json 2>/dev/null
  python - <<PY
import json,re,collections
d=json.loads(open('gpurun_out/b_$v.json').read().strip().splitlines()[-1])
agg=collections.defaultdict(lambda:[0,0.0])
for l in open('gpurun_out/l_$v.jsonl'):
    r=json.loads(l)
    if re.search(r'$PAT', r['label']):
        k=(r['label'], r.get('N'), r.get('C')); agg[k][0]+=1; agg[k][1]+=r['us']
print('$VAR=$v', d['ms_per_step'], {str(k):(n,round(t/n,1)) for k,(n,t) in agg.items()})
PY
done
