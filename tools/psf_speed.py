"""Ad-hoc: PSF pixel-grid stage rate for F frames of n x n stamps: python tools/psf_speed.py F S n iters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd import _lib
if os.environ.get('LCMI_DBG_LIB'): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['LCMI_DBG_LIB'])
from lightcurver_amd.psf_batch import PsfBatch
from lightcurver_amd.synthetic import make_psf_dataset
F, S, n, iters = [int(x) for x in sys.argv[1:5]]
ds = make_psf_dataset(F=F, S=S, n=n, ss=2, seed=103)
ctx = _lib.Context(0)
w = (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)
b = PsfBatch(ds['data'], w, 2, ctx)
g = ds['fwhm_guess']; f0 = np.sqrt(np.maximum(g * g - 1, 1.0))
b.set_moffat(np.stack([f0, f0, np.zeros(F), np.full(F, 2.5)], -1))
st = np.zeros((F, S, 4), np.float32); st[..., 0] = (ds['data'] * ds['masks']).sum((-1, -2)); b.set_stars(st)
b.set_grid(None); b.propagate_noise(); b.set_regularization(None, 1.0, 1.0)
b.run_adabelief(10, init_learning_rate=1e-4); ctx.synchronize()
ctx.timer_start(); b.run_adabelief(iters, init_learning_rate=1e-4); ms = ctx.timer_stop()
h = b.loss_history()
print(f'F={F} S={S} n={n}: {ms / iters * 1e3:.1f} us/iter, {F * S * iters / (ms * 1e-3):.3e} cutouts/s, loss finite {np.isfinite(h).all()} last {float(np.asarray(h)[..., -1].sum())!r}, single_wg={os.environ.get("LCMI_PSF_SINGLE_WG")}')
