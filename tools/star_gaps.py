import csv, glob, statistics as st, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
byq = collections.defaultdict(list)
for r in tr:
    if 'joint_ps_kernel' in r['Kernel_Name'] or 'joint_update_groups' in r['Kernel_Name']:
        byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    rs = rs[len(rs)//3: 2*len(rs)//3]
    d = collections.defaultdict(list); gaps = collections.defaultdict(list)
    for a, b in zip(rs[:-1], rs[1:]):
        gaps[a['Kernel_Name'][:24] + '->' + b['Kernel_Name'][:24]].append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
    for r in rs: d[r['Kernel_Name'][:40]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('queue', q, {k: round(st.median(v), 2) for k, v in d.items()}, {k: round(st.median(v), 2) for k, v in gaps.items()})
