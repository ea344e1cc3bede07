"""Soak run of the two-workgroup PSF kernel: 30 000 iterations in one launch and 20 launches of 1000, once with the same-XCD
hand-off (default) and once with the write-through hand-off everywhere (LCMI_PSF_XCD_FAST=0): same bits, no fall-back taken."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd import _lib
from lightcurver_amd.psf_batch import PsfBatch
from lightcurver_amd.synthetic import CONFIGS, make_psf_dataset
cfg = dict(CONFIGS['C2']); cfg.pop('kind')
ds = make_psf_dataset(**cfg)
F, S, n, ss = cfg['F'], cfg['S'], cfg['n'], cfg['ss']
ctx = _lib.Context(0)
w = (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)
out = []
for env in ({}, {'LCMI_PSF_XCD_FAST': '0'}):
    os.environ.update(env)
    try:
        b = PsfBatch(ds['data'], w, ss, ctx)
        g = ds['fwhm_guess']; f0 = np.sqrt(np.maximum(g * g - 1, 1.0))
        b.set_moffat(np.stack([f0, f0, np.zeros(F), np.full(F, 2.5)], -1))
        st = np.zeros((F, S, 4), np.float32); st[..., 0] = (ds['data'] * ds['masks']).sum((-1, -2)); b.set_stars(st)
        b.set_grid(None); b.fit_moffat(50); b.propagate_noise(); b.set_regularization(None, 1.0, 1.0)
        t0 = time.perf_counter()
        b.run_adabelief(30000, init_learning_rate=1e-4); ctx.synchronize()
        print(env, '30000 iterations in one launch:', time.perf_counter() - t0, 's')
        for k in range(20):
            b.run_adabelief(1000, init_learning_rate=1e-4)
        h = b.loss_history()
        print('total iterations', b.iterations_done, 'finite', np.isfinite(h).all(), 'loss first/last', h[0, 0], h[0, -1],
              'chi2', np.median(b.results()['chi2']), 'fall-backs', b.split_fallbacks)
        out.append((h.copy(), b.get_grid(), b.get_stars()))
        b.close()
    finally:
        for k in env:
            os.environ.pop(k, None)
same = all(np.array_equal(a, c) for a, c in zip(out[0], out[1]))
print('bit-identical histories, grids and star parameters over 50 000 iterations:', same)
sys.exit(0 if same else 1)
