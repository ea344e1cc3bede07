"""Ad-hoc: what the sharded loop (step_local / all-reduce / step_update) costs against the one-GPU device loop, at world size 1
(no waiting in the all-reduce): python tools/shard_overhead.py E n M iterations"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29533')
dist.init_process_group('gloo', rank=0, world_size=1)
from lightcurver_amd import _lib
from lightcurver_amd.joint import JointFit
from lightcurver_amd.distributed import PeerGroup, ShardedJointOptimizer
from lightcurver_amd.synthetic import make_roi_dataset
E = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M = int(sys.argv[3]) if len(sys.argv) > 3 else 2
IT = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
ctx = _lib.Context(0)
def mk():
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64)**2, ds['psf'], 2, M, ctx)
    p = dict(ds['truth']); p['a'] = np.asarray(p['a'])*0.9
    j.set_params(**p)
    j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
    j.set_free(['a','c_x','c_y','dx','dy','mean','h'])
    return j
ab = dict(init_learning_rate=1e-4, schedule_learning_rate=False)
j = mk(); j.run_adabelief(10, **ab); ctx.synchronize(); t0=time.perf_counter(); j.run_adabelief(IT, **ab); ctx.synchronize(); print('unsharded', (time.perf_counter()-t0)*1e6/IT, 'us/iter'); j.close()
j = mk(); peer = PeerGroup(j); opt = ShardedJointOptimizer(j, None, peer=peer); opt.run(10, **ab); ctx.synchronize(); t0=time.perf_counter(); opt.run(IT, **ab); ctx.synchronize(); print('sharded loop, peer kernel, world 1:', (time.perf_counter()-t0)*1e6/IT, 'us/iter'); peer.close(); j.close()
j = mk(); opt = ShardedJointOptimizer(j, None); opt.run(10, **ab); ctx.synchronize(); t0=time.perf_counter(); opt.run(IT, **ab); ctx.synchronize(); print('sharded loop, gloo world 1 (no-op reduce):', (time.perf_counter()-t0)*1e6/IT, 'us/iter'); j.close()
