#!/bin/bash
# Runs on the GPU box: HBM traffic of one joint-fit iteration (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, no tracing;
# units and gfx950 corrections as in bench.py / MI355X_MICROARCH.md).  usage: bash tools/pmc_hbm_joint.sh E n M [iters]
E=${1:-125}; n=${2:-128}; M=${3:-4}; IT=${4:-20}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_hbm_joint_${E}_${n}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  # (counter collection serialises the two streams: the update must wait for the regulariser by an event, not by polling)
  LCMI_EVENT_SYNC=1 LCMI_PTS=0.01 LCMI_FU=10 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $REPO/tools/joint_speed.py $E $n $M $IT > $OUT/$c.log 2>&1
done
cd $REPO
python3 - <<PY
import csv, glob, collections, json
E, n, IT = $E, $n, $IT
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c: continue
            k = r['Kernel_Name'].split('(')[0].replace('void lc::', '').replace('lc::', '')
            tot[k][c] += float(r['Counter_Value'])
            cnt[k][c] += 1
out = {}
grand = 0.0
for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
    if not any(s in k for s in ('joint_epoch', 'joint_reduce_update', 'mreg_')): continue
    calls = max(cnt[k].values())
    if calls < IT: continue  # (the kernels of the iterations, not those of the set-up)
    # KiB units; FETCH_SIZE counts half the bytes on gfx950 (x2)
    fetch = 2.0 * tot[k]['FETCH_SIZE'] * 1024 / calls
    write = tot[k]['WRITE_SIZE'] * 1024 / calls
    out[k] = {'launches': calls, 'fetch_MB_per_launch': round(fetch / 1e6, 2), 'write_MB_per_launch': round(write / 1e6, 2)}
    grand += fetch + write
out['_sum_over_kernels_MB_per_iteration'] = round(grand / 1e6, 1)
out['_config'] = {'E': E, 'n': n, 'iterations': IT, 'note': 'one launch of every kernel per iteration; FETCH_SIZE x2 (gfx950), KiB units; L2-miss traffic as MI355X_MICROARCH.md prescribes'}
print(json.dumps(out, indent=1))
json.dump(out, open('$OUT/summary.json', 'w'), indent=1)
PY
