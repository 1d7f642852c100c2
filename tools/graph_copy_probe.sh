#!/bin/bash
# counts copyBuffer launches per graph replay for graphs of 10 / 100 / 400 kernel nodes (20 replays each)
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/graph_probe; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for N in 10 100 400; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/n$N -- python3 $ROOT/tools/graph_copy_probe.py $N 20 > $OUT/n$N.log 2>&1 || { tail -5 $OUT/n$N.log; exit 1; }
  F=$(find $OUT/n$N -name '*kernel_trace.csv' | head -1)
  echo "nodes=$N: copyBuffer launches = $(grep -c copyBuffer $F), elementwise = $(grep -c elementwise $F), total rows = $(wc -l < $F)"
done
rm -rf $OUT/n10 $OUT/n100 $OUT/n400
