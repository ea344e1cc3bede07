import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from lightcurver_amd import _lib
from lightcurver_amd.joint import JointFit
from lightcurver_amd.synthetic import make_roi_dataset
ctx = _lib.Context(0)
E, M, n = 8, 2, 64
ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
def fit(T, lams, fused):
    os.environ['LCMI_REG_FUSED'] = fused
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
    p = {k: np.asarray(v, dtype=np.float64) for k, v in ds['truth'].items()}
    p['a'] = 0.9 * p['a']
    j.set_params(**p)
    W = j.propagate_noise()
    j.set_loss(W=W, **lams)
    j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
    j.run_adabelief(T, init_learning_rate=1e-4, schedule_learning_rate=False)
    h = np.asarray(j.loss_history(), dtype=np.float64)
    j.close()
    return h
base = dict(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
for name, ov in [('all', {}), ('no pts', dict(lam_pts_source=0.0)), ('no pos', dict(lam_positivity=0.0)), ('no hf', dict(lam_hf=0.0)), ('no scales', dict(lam_scales=0.0)),
                 ('only pts+hf', dict(lam_scales=0.0, lam_positivity=0.0))]:
    l = dict(base); l.update(ov)
    a = fit(4, l, '1'); b = fit(4, l, '0')
    print(name, a - b)
