#!/bin/bash
# Runs on the GPU box: per-iteration time of the 64 x 64 joint fit at the epoch counts a sharded C4 leaves per GPU, in the
# one-workgroup (LDS spectrum) form and in the split forms.  usage: bash tools/r3_split_sweep.sh <outfile>
OUT=${1:-gpurun_out/split_sweep.txt}
export LCMI_PTS=0.01 LCMI_FU=10
: > $OUT
for E in 25 50 100; do
  echo "== E=$E one workgroup per epoch" >> $OUT
  LCMI_N128_SPLIT=0 python3 tools/joint_speed.py $E 64 2 500 2>&1 | grep 'us/iter' >> $OUT
  for NW in 4; do
    echo "== E=$E split NW=$NW default parts" >> $OUT
    LCMI_N128_SPLIT=1 LCMI_N128_NW=$NW python3 tools/joint_speed.py $E 64 2 500 2>&1 | grep 'us/iter' >> $OUT
  done
done
