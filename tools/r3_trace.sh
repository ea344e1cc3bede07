#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of a joint fit.  usage: bash tools/r3_trace.sh <label> E n M iters [env assignments...]
LABEL=$1; E=$2; n=$3; M=$4; IT=$5; shift 5
REPO=$(pwd)
OUT=$REPO/gpurun_out/trace_$LABEL
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp LCMI_PTS=0.01 LCMI_FU=10
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/tools/joint_speed.py $E $n $M $IT > $OUT/run.log 2>&1
cd $REPO
f=$(find $OUT/t -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -25 $OUT/kernel_stats.csv | cut -c1-200
grep 'us/iter' $OUT/run.log
