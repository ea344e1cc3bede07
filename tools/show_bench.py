"""Prints the bench line in short form: python tools/show_bench.py bench_line.json"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get('roofline', {})
print(f"headline {d['value']:.4g} {d['unit']}  {d['ms_per_step']:.3f} ms/step  roofline.frac {r.get('frac')}  traffic {r.get('traffic')}")
c = d.get('cpu_baseline', {})
print(f"cpu_baseline {c.get('value')} {c.get('unit')} on {c.get('cores')} cores ({c.get('kind')})")
for w in d['config'].get('other_workloads', []):
    rr = w.get('roofline', {})
    print(f"  {w['workload'][:70]:70s} {w.get('us_per_iteration', float('nan')):8.2f} us/it  frac {rr.get('frac')}  traffic {rr.get('traffic')}")
