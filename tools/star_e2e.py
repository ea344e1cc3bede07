"""Wall-clock profile of one star's joint fit through the restated step function (E epochs of n x n)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd.synthetic import make_roi_dataset
from lightcurver_amd.processes.star_photometry import do_one_star_forward_modelling
E, n = (int(x) for x in sys.argv[1:3]) if len(sys.argv) > 2 else (100, 32)
ds = make_roi_dataset(E=E, M=1, n=n, ss=2, seed=3, with_background=False)
for rep in range(2):
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    out = do_one_star_forward_modelling(ds['data'].astype(np.float64).copy(), ds['noisemap'].astype(np.float64).copy(), ds['psf'], 2, n_iter=2000)
    pr.disable(); print(f'total {time.perf_counter() - t0:.3f} s; chi2 {out["chi2"]:.3f}')
pstats.Stats(pr).sort_stats('cumulative').print_stats(16)
