"""Wall-clock profile of one C4-sized two-stage ROI fit through the restated step function."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd.synthetic import make_roi_dataset
from lightcurver_amd.processes.roi_modelling import model_roi_cutouts
E, n, M = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (200, 64, 2)
ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
off = (n - 1) / 2.0
xs = np.asarray(ds['truth']['c_x']) + off
ys = np.asarray(ds['truth']['c_y']) + off
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    out = model_roi_cutouts(ds['data'].copy(), ds['noisemap'].copy(), ds['psf'], 2, xs, ys)
    pr.disable()
    print(f'total {time.perf_counter() - t0:.3f} s; final loss {out["loss_history"][-1]:.4g}')
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(18)
