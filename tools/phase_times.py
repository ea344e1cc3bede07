"""Per-phase durations of the phased joint epoch launches from a rocprofv3 kernel_stats.csv (diagnostic)."""
import csv
import re
import sys

for f in sys.argv[1:]:
    rows = {}
    for r in csv.DictReader(open(f)):
        m = re.search(r'joint_epoch_kernel<lc::JointCfg<(\d+), \d, \d+, \d+, \w+, \d+, (\w+)>, false, (\d)', r['Name'])
        if m:
            rows['phase %s%s' % (m.group(3), ' tile' if m.group(2) == 'true' else '')] = float(r['AverageNs']) / 1e3
        elif 'joint_epoch_finish' in r['Name']:
            rows['finish'] = float(r['AverageNs']) / 1e3
    print(f.split('/')[-3], ' '.join('%s=%.1f' % kv for kv in sorted(rows.items())), 'sum=%.1f us' % sum(rows.values()))
