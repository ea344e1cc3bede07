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
LCMI_PTS=0.01 LCMI_FU=10 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5full -- python3 $REPO/tools/joint_speed.py 1000 128 4 30 > $OUT/c5full.log 2>&1
echo "c5 (1000 epochs) trace done"
LCMI_PTS=0.01 LCMI_FU=10 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4shard -- python3 $REPO/tools/joint_speed.py 25 64 2 500 > $OUT/c4shard.log 2>&1
echo "c4 shard trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shardloop -- python3 $REPO/tools/shard_overhead.py 25 64 2 300 > $OUT/shardloop.log 2>&1
echo "sharded loop (world 1) trace done"
cd $REPO
for k in bench c4 c5 c5full c4shard shardloop; do
  f=$(find $OUT/$k -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $DST/${LABEL}_${k}_kernel_stats_raw.csv
done
tail -1 $OUT/bench.log > $DST/${LABEL}_bench_line_under_rocprof.json
grep -E 'us/iter|^loss' $OUT/c4.log > $DST/${LABEL}_c4_joint_speed.txt
grep -E 'us/iter|^loss' $OUT/c5.log > $DST/${LABEL}_c5_joint_speed.txt
grep -E 'us/iter|^loss' $OUT/c5full.log > $DST/${LABEL}_c5full_joint_speed.txt
grep -E 'us/iter|^loss' $OUT/c4shard.log > $DST/${LABEL}_c4shard_joint_speed.txt
grep -E 'us/iter' $OUT/shardloop.log > $DST/${LABEL}_shardloop_under_rocprof.txt
# the same fits (and 200 epochs of 128 x 128) without the profiler attached (its tracing costs a few us per launch)
for cfg in "200 64 2 1000" "25 64 2 1000" "125 128 4 150" "200 128 4 100" "1000 128 4 30"; do LCMI_PTS=0.01 LCMI_FU=10 python3 $REPO/tools/joint_speed.py $cfg 2>&1 | grep 'us/iter'; done > $DST/${LABEL}_joint_speed_no_profiler.txt
head -12 $DST/${LABEL}_bench_kernel_stats_raw.csv
for cfg in "25 64 2 500" "125 128 4 150" "200 64 2 500"; do echo "== $cfg"; python3 $REPO/tools/shard_overhead.py $cfg 2>&1 | grep 'us/iter'; done > $DST/${LABEL}_sharded_loop_world1.txt
