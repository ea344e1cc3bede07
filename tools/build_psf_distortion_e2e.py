"""Wall clock of build_psf_batch(field_distortion=True) on a C2-sized list of frames (the mode the reference's integration test runs)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd.synthetic import CONFIGS, make_psf_dataset
from lightcurver_amd.starred.procedures.psf_routines import build_psf_batch
cfg = dict(CONFIGS['C2']); cfg.pop('kind')
if len(sys.argv) > 1: cfg['F'] = int(sys.argv[1])
ds = make_psf_dataset(**cfg)
F, S = cfg['F'], cfg['S']
rng = np.random.default_rng(3)
imgs = [ds['data'][f] for f in range(F)]; nois = [ds['noisemap'][f] for f in range(F)]; masks = [ds['masks'][f] for f in range(F)]
coords = [rng.uniform(-0.5, 0.5, (S, 2)) for _ in range(F)]
for rep in range(2):
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    res = build_psf_batch(imgs, nois, 2, masks=masks, n_iter_analytic=100, n_iter_adabelief=3000, guess_method_star_position='center',
                          guess_fwhm_pixels=ds['fwhm_guess'], field_distortion=True, stamp_coordinates=coords)
    pr.disable(); print(f'total {time.perf_counter() - t0:.3f} s; median chi2 {np.median([r["chi2"] for r in res]):.3f}')
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
