#!/bin/bash
# Runs on the GPU box: kernel trace + the two PMC passes of the default bench command, then the summaries.
# usage: bash tools/profile_round.sh <label>
set -e
LABEL=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$LABEL
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-joint > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-joint > $OUT/write.log 2>&1
echo "write done"
cd $REPO
python3 tools/parse_rocprof.py $OUT/trace $OUT/fetch $OUT/write $LABEL
mkdir -p gpurun_out/profiles_$LABEL
cp profiles/${LABEL}_kernel_stats.csv profiles/${LABEL}_pmc_summary.json profiles/pmc_summary.json gpurun_out/profiles_$LABEL/
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f gpurun_out/profiles_$LABEL/${LABEL}_rocprofv3_kernel_stats_raw.csv
tail -1 $OUT/trace.log > gpurun_out/profiles_$LABEL/${LABEL}_bench_line_under_rocprof.json
