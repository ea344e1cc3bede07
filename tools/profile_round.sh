#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel traces behind the bench line (the HBM counters are measured by bench.py itself,
# in its own --pmc passes).  usage: bash tools/profile_round.sh <label>
set -e
LABEL=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$LABEL
DST=$REPO/gpurun_out/profiles_$LABEL
mkdir -p $OUT $DST
export TMPDIR=/tmp
cd /tmp
# the bench command itself (fewer steps; the timed kernel and its launch shape are the same)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-traffic --no-cpu-baseline > $OUT/bench.log 2>&1
echo "bench trace done"
LCMI_PTS=0.01 LCMI_FU=10 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- python3 $REPO/tools/joint_speed.py 200 64 2 300 > $OUT/c4.log 2>&1
echo "c4 trace done"
LCMI_PTS=0.01 LCMI_FU=10 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -- python3 $REPO/tools/joint_speed.py 125 128 4 100 > $OUT/c5.log 2>&1
echo "c5 trace done"
cd $REPO
for k in bench c4 c5; do
  f=$(find $OUT/$k -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $DST/${LABEL}_${k}_kernel_stats_raw.csv
done
tail -1 $OUT/bench.log > $DST/${LABEL}_bench_line_under_rocprof.json
tail -3 $OUT/c4.log > $DST/${LABEL}_c4_joint_speed.txt
tail -3 $OUT/c5.log > $DST/${LABEL}_c5_joint_speed.txt
head -12 $DST/${LABEL}_bench_kernel_stats_raw.csv
