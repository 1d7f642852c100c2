#!/bin/bash
# Same-box A/B of one environment switch: tools/ab_bench.sh VAR v1 v2 [v1 again ...] -> ms per step of the default bench for each value, in order.
# Run through gpurun from the repo root; writes gpurun_out/ab_<VAR>_<value>_<k>.json.
VAR=$1; shift
k=0
for v in "$@"; do
  k=$((k+1))
  out=gpurun_out/ab_${VAR}_${v}_${k}.json
  env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-events > $out 2> gpurun_out/ab_${VAR}.err || { tail -5 gpurun_out/ab_${VAR}.err; exit 1; }
  python - "$out" "$VAR=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f'{sys.argv[2]:32s} {d["ms_per_step"]:.3f} ms/step  {d["value"] / 1e6:.3f} M frames/s')
PY
done
