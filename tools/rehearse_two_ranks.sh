#!/bin/bash
# Rehearsal of the N > 1 bench path on a ONE-GPU box: two ranks share cuda:0, gradients travel over gloo (DX_BENCH_REHEARSAL=1).
# Run through gpurun from the repo root; prints the one JSON line's key fields.
DX_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/rehearsal2.json 2> gpurun_out/rehearsal2.err || { tail -5 gpurun_out/rehearsal2.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/rehearsal2.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['n_gpus'], d['config']['parallelism'], d['config']['launch'][:60], d['exchange']['exposed_fraction'])
PY
