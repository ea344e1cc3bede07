#!/bin/bash
# Runs on the GPU box: SQ wave-time breakdown of the timed PSF kernel (one --pmc pass, no tracing).
# usage: bash tools/pmc_sq.sh <label>
LABEL=${1:-sq}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$LABEL
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --pmc-child > $OUT/p1.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/p1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'psf_fit_kernel' not in r['Kernel_Name']: continue
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if us < 5000: continue
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
tot = {k: sum(v) / len(v) for k, v in acc.items()}
for k, v in sorted(tot.items()): print(f'{k:28s} {v:.4g}')
wc = tot.get('SQ_WAVE_CYCLES', 1)
for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS'):
    if k in tot: print(f'{k}/WAVE_CYCLES = {tot[k] / wc:.3f}')
PY
