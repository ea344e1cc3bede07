"""Merge the SQ wave-time counters of tools/pmc_sq_joint.sh runs (gpurun_out/pmc_joint_<E>_<n>/) into
profiles/r02_sq_wave_time_breakdown.json under '<label>: <kernel>' keys.  usage: python tools/sq_merge.py label E n"""
import collections, csv, glob, json, os, sys
label, E, n = sys.argv[1:4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, f'gpurun_out/pmc_joint_{E}_{n}/p1/**/*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
path = os.path.join(root, 'profiles/r02_sq_wave_time_breakdown.json')
out = json.load(open(path))
for k, d in acc.items():
    tot = {c: sum(v) / len(v) for c, v in d.items()}
    wc = tot.get('SQ_WAVE_CYCLES', 0)
    if wc < 1e6:
        continue
    e = {c: round(tot[c] / wc, 4) for c in sorted(tot) if c != 'SQ_WAVE_CYCLES'}
    e['launches'] = len(next(iter(d.values())))
    e['SQ_INSTS_VALU_per_launch'] = round(tot.get('SQ_INSTS_VALU', 0))
    out[f'{label}: {k}'] = e
json.dump(out, open(path, 'w'), indent=1)
print('merged', [k for k in out if k.startswith(label)])
