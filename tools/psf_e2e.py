"""Wall-clock split of one C2-sized build_psf batch: Moffat stage, noise propagation, pixel-grid stage, outputs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd import _lib
from lightcurver_amd.psf_batch import PsfBatch
from lightcurver_amd.synthetic import CONFIGS, make_psf_dataset
cfg = dict(CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'C2']); cfg.pop('kind')
ds = make_psf_dataset(**cfg)
F, S, n, ss = cfg['F'], cfg['S'], cfg['n'], cfg['ss']
ctx = _lib.Context(0)
weight = (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)
for rep in range(2):
    t0 = time.perf_counter()
    b = PsfBatch(ds['data'], weight, ss, ctx)
    g = ds['fwhm_guess']; f0 = np.sqrt(np.maximum(g * g - (2.0 / ss) ** 2, 1.0))
    b.set_moffat(np.stack([f0, f0, np.zeros(F), np.full(F, 2.5)], axis=-1))
    stars = np.zeros((F, S, 4), np.float32); stars[..., 0] = (ds['data'] * ds['masks']).sum(axis=(-1, -2))
    b.set_stars(stars); b.set_grid(None); ctx.synchronize()
    t1 = time.perf_counter(); b.fit_moffat(100); ctx.synchronize()
    t2 = time.perf_counter(); b.propagate_noise(); ctx.synchronize()
    t3 = time.perf_counter(); b.set_regularization(None, 1.0, 1.0); b.run_adabelief(3000, init_learning_rate=1e-4); ctx.synchronize()
    t4 = time.perf_counter(); res = b.results(); h = b.loss_history()
    t5 = time.perf_counter()
    print(f'create+upload {t1-t0:.3f}s  moffat(100) {t2-t1:.3f}s  propagate_noise {t3-t2:.3f}s  adabelief(3000) {t4-t3:.3f}s  results {t5-t4:.3f}s  total {t5-t0:.3f}s')
