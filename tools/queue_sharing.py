"""What happens to the device loop when the runtime maps the two streams of a fit onto ONE hardware queue (few queues, many live
objects): the probe of csrc/joint_fit.hip (probe_streams) must notice and keep the event form; numbers and speed either way.
python tools/queue_sharing.py [live objects]      (GPU_MAX_HW_QUEUES=<n> in the environment limits the runtime's queues)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['LCMI_DEBUG_STREAMS'] = '1'
import numpy as np
from lightcurver_amd import _lib
from lightcurver_amd.joint import JointFit
from lightcurver_amd.synthetic import make_roi_dataset
live = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = _lib.Context(0)
ds = make_roi_dataset(E=25, M=2, n=64, ss=2, seed=104)
objs = []
for k in range(live):
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, 2, ctx)
    p = dict(ds['truth']); p['a'] = np.asarray(p['a']) * 0.9
    j.set_params(**p)
    W = j.propagate_noise()
    j.set_loss(W=W, lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
    j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
    objs.append(j)
for k, j in enumerate(objs):
    j.run_adabelief(20, init_learning_rate=1e-4, schedule_learning_rate=False); ctx.synchronize()
    t0 = time.perf_counter()
    j.run_adabelief(500, init_learning_rate=1e-4, schedule_learning_rate=False); ctx.synchronize()
    h = j.loss_history()
    print(f'object {k}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us/iter, loss {h[0]:.2f} -> {h[-1]:.4f}', flush=True)
for j in objs:
    j.close()
