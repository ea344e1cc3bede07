"""Ad-hoc: where the wall time of do_many_stars_forward_modelling goes (host steps against the device loop):
python tools/star_batch_profile.py [G E n iters]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd import _lib
from lightcurver_amd.processes.star_photometry import do_many_stars_forward_modelling, do_one_star_forward_modelling
from lightcurver_amd.synthetic import make_roi_dataset
G, E, n, iters = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (30, 100, 32, 2000)
base = make_roi_dataset(E=E, M=1, n=n, ss=2, seed=77, with_background=False)
rng = np.random.default_rng(5)
def stacks():
    out = []
    for g in range(G):
        f = rng.uniform(0.3, 3.0)
        d = (base['data'].astype(np.float64) * f) * base['scale']
        nm = base['noisemap'].astype(np.float64) * np.sqrt(f) * base['scale']
        out.append((d, nm, base['psf']))
    return out
do_many_stars_forward_modelling(stacks()[:2], 2, n_iter=5)   # warm-up
_lib.Context(0).synchronize()
s = stacks()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
do_many_stars_forward_modelling(s, 2, n_iter=iters)
pr.disable(); print('batch wall', time.perf_counter() - t0)
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
s = stacks()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for d, nm, p in s:
    do_one_star_forward_modelling(d, nm, p, 2, n_iter=iters, starlet_global_background=False)
pr.disable(); print('loop wall', time.perf_counter() - t0)
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
