#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   bench kernel table (events)  ->  rocprofv3 kernel trace  ->  rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs)
# and writes the summaries under gpurun_out/profiles_<tag>/ (copy the ones to be judged into profiles/).
set -o pipefail
TAG=${1:-r02}
PREC=${2:-bf16}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 --precision $PREC --no-cpu-baseline --no-extras --kernel-table $OUT/${TAG}_${PREC}_kernel_table.json > $OUT/${TAG}_bench_${PREC}.json 2> $OUT/bench.err || exit 1
echo "bench done: $(head -c 200 $OUT/${TAG}_bench_${PREC}.json)"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 2 --precision $PREC --no-cpu-baseline --no-extras --no-kernel-events > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --precision $PREC --no-cpu-baseline --no-extras --no-kernel-events > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --precision $PREC --no-cpu-baseline --no-extras --no-kernel-events > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
echo "write done"
cd $ROOT
python3 tools/summarize_profile.py pmc $OUT/fetch $OUT/write $PREC $OUT/${TAG}_pmc_traffic.json > $OUT/pmc_summary.log 2>&1 || { tail -5 $OUT/pmc_summary.log; exit 1; }
python3 tools/summarize_profile.py stats $OUT/trace 8 $OUT/${TAG}_${PREC}_kernel_stats.csv $OUT/${TAG}_${PREC}_kernel_table.json $OUT/${TAG}_pmc_traffic.json || exit 1
# the raw traces are large: keep only the summaries
rm -rf $OUT/trace $OUT/fetch $OUT/write
ls -la $OUT
