"""Summarise rocprofv3 CSV output of `bench.py` runs into profiles/ (round-labelled).
   python tools/parse_rocprof.py <trace_dir> <fetch_dir> <write_dir> <label>
The timed launches of the persistent PSF-fit kernel (ITERS_PER_STEP AdaBelief iterations each) are told
apart from the one-iteration evaluation launches of the untimed setup by their duration (> 1 ms)."""
import csv, glob, json, os, sys
from collections import defaultdict

trace_dir, fetch_dir, write_dir, label = sys.argv[1:5]
LONG_US = 1000.0


def rows(d, pattern):
    for f in glob.glob(os.path.join(d, '**', pattern), recursive=True):
        yield from csv.DictReader(open(f))


def name(r):
    n = r['Kernel_Name']
    for key in ('psf_fit_kernel', 'joint_epoch_kernel', 'joint_update_kernel', 'joint_reduce_kernel', 'moffat_raster_kernel',
                'moffat_grad_kernel', 'psf_finalize_kernel', 'psf_residual_kernel'):
        if key in n:
            return key
    return n.split('(')[0][:60]


dur = defaultdict(list)
for r in rows(trace_dir, '*kernel_trace.csv'):
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = name(r)
    if k == 'psf_fit_kernel':
        if int(r.get('Workgroup_Size_X', 512)) == 1024:
            k += ' [64x64 stamps: C3-shard secondary figure]'
        else:
            k += ' [timed launch, 100 iterations]' if us > LONG_US else ' [1-iteration evaluation, untimed setup]'
    dur[k].append(us)
total = sum(sum(v) for v in dur.values())
ktab = [dict(kernel=k, calls=len(v), total_us=round(sum(v), 1), avg_us=round(sum(v) / len(v), 2), min_us=round(min(v), 2),
             max_us=round(max(v), 2), pct=round(100 * sum(v) / total, 2)) for k, v in dur.items()]
ktab.sort(key=lambda r: -r['total_us'])


def counter(d, cname):
    vals = defaultdict(list)
    for r in rows(d, '*counter_collection.csv'):
        if r['Counter_Name'] != cname:
            continue
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        k = name(r)
        if k == 'psf_fit_kernel' and us > LONG_US and int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 512))) != 1024:
            vals['psf_fit_kernel'].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


fetch, nf = counter(fetch_dir, 'FETCH_SIZE')
write, nw = counter(write_dir, 'WRITE_SIZE')
pmc = {}
for k in fetch:
    f_kb, w_kb = fetch[k], write.get(k, 0.0)
    pmc[k] = dict(FETCH_SIZE_KB_per_launch=f_kb, WRITE_SIZE_KB_per_launch=w_kb, launches_fetch=nf[k], launches_write=nw.get(k, 0),
                  # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> x2;
                  # WRITE_SIZE is exact for 16-byte stores.  Units are KiB.
                  hbm_bytes_per_launch=(2.0 * f_kb + w_kb) * 1024.0)
os.makedirs('profiles', exist_ok=True)
with open(os.path.join('profiles', f'{label}_kernel_stats.csv'), 'w', newline='') as fh:
    w = csv.DictWriter(fh, fieldnames=list(ktab[0].keys()))
    w.writeheader()
    w.writerows(ktab)
json.dump(pmc, open(os.path.join('profiles', f'{label}_pmc_summary.json'), 'w'), indent=1)
json.dump(pmc, open(os.path.join('profiles', 'pmc_summary.json'), 'w'), indent=1)
for r in ktab[:8]:
    print(r)
print(json.dumps(pmc, indent=1))
