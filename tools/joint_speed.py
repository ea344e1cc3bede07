"""Ad-hoc timing of the joint fit (not the bench contract): python tools/joint_speed.py E n M iters"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd import _lib
if os.environ.get('LCMI_DBG_LIB'): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['LCMI_DBG_LIB'])
from lightcurver_amd.joint import JointFit
from lightcurver_amd.synthetic import make_roi_dataset
E, n, M, iters = [int(x) for x in sys.argv[1:5]]
if os.environ.get("LCMI_DEBUG_GLOBAL"): _lib.lib().lc_joint_set_debug_global(1)
with_h = (len(sys.argv) < 6 or sys.argv[5] != 'noh')
t0 = time.time(); ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104); print('synth', time.time() - t0)
ctx = _lib.Context(0)
t0 = time.time(); j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx); print('create', time.time() - t0)
p = dict(ds['truth']); p['a'] = p['a'] * 0.9
if not with_h: p['h'] = np.zeros_like(p['h'])
j.set_params(**p)
t0 = time.time(); W = j.propagate_noise(); print('propagate_noise', time.time() - t0)
lam = 0.0 if os.environ.get('LCMI_LAM0') else 1.0   # LCMI_LAM0=1: no regulariser at all (the epoch path alone)
j.set_loss(W=W, lam_scales=lam, lam_hf=lam, lam_positivity=100.0 * lam, lam_pts_source=float(os.environ.get('LCMI_PTS', '0')), lam_flux_uniformity=float(os.environ.get('LCMI_FU', '0')))
free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean'] + (['h'] if with_h else [])
j.set_free(free)
j.run_adabelief(5, init_learning_rate=1e-4, schedule_learning_rate=False); ctx.synchronize()
if os.environ.get('LCMI_PHIST'): print('param history rows of', j.param_history_begin(iters + 8), 'floats, recorded on the device')
ctx.timer_start(); t0 = time.time()
j.run_adabelief(iters, init_learning_rate=1e-4, schedule_learning_rate=False)
ms = ctx.timer_stop(); wall = time.time() - t0
print(f'E={E} n={n} M={M} with_h={with_h}: {ms / iters * 1e3:.1f} us/iter (device), wall {wall / iters * 1e6:.1f} us/iter, '
      f'{E * iters / (ms * 1e-3):.3e} cutout-iterations/s')
h = j.loss_history(); print('loss', h[0], h[-1])
