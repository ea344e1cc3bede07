"""Median kernel durations and gaps of the main-stream kernels of a joint iteration from a rocprofv3 kernel trace:
python tools/iter_gaps.py <dir with *kernel_trace.csv>"""
import csv, glob, statistics as st, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ep, up, g1, g2, it = [], [], [], [], []
chain = {}
last_upd_end = last_ep_end = None
for r in tr:
    n, s, e = r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'joint_epoch_kernel' in n and 'true, 0>' not in n:
        if last_upd_end: g1.append((s - last_upd_end) / 1e3)
        last_ep_end = e; ep.append((e - s) / 1e3)
    elif 'reduce_update' in n or 'joint_update_gm' in n:
        if last_ep_end: g2.append((s - last_ep_end) / 1e3)
        if last_upd_end: it.append((e - last_upd_end) / 1e3)
        last_upd_end = e; up.append((e - s) / 1e3)
    elif 'mreg_' in n:
        k = n.split('(')[0][-28:]
        chain.setdefault(k, []).append(((e - s) / 1e3, (s - last_upd_end) / 1e3 if last_upd_end else 0.0))
med = lambda x: st.median(x[len(x) // 4:]) if x else float('nan')
print(f'gap update_end -> epoch_start {med(g1):.2f}  epoch {med(ep):.2f}  gap epoch_end -> update_start {med(g2):.2f}  update {med(up):.2f}  iteration {med(it):.2f}')
for k, v in chain.items():
    print(f'   {k:30s} dur {med([a for a, _ in v]):6.2f}  starts {med([b for _, b in v]):6.2f} us after the previous update ended')
