"""Ad-hoc: the field_distortion=True pixel stage alone (bench.py's distortion_workload), for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lightcurver_amd import _lib
ctx = _lib.Context(0)
r = bench.distortion_workload(ctx, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 100)
print(r['workload'], r['value'], r['us_per_iteration'])
