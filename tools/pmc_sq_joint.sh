#!/bin/bash
# Runs on the GPU box: SQ wave-time breakdown of the joint epoch kernel (one --pmc pass).  usage: bash tools/pmc_sq_joint.sh E n M
E=${1:-200}; n=${2:-64}; M=${3:-2}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_joint_${E}_${n}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
LCMI_EVENT_SYNC=1 LCMI_PTS=0.01 LCMI_FU=10 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- python3 $REPO/tools/joint_speed.py $E $n $M 100 > $OUT/p1.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    tot = {c: sum(v) / len(v) for c, v in d.items()}
    wc = tot.get('SQ_WAVE_CYCLES', 0)
    if wc < 1e6: continue
    print(k, 'launches', len(next(iter(d.values()))))
    for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_LDS_BANK_CONFLICT'):
        if c in tot: print(f'   {c}/WAVE_CYCLES = {tot[c] / wc:.3f}')
PY
