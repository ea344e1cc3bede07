"""Ad-hoc timing of the batched star photometry: python tools/star_batch_speed.py G E n iters"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd import _lib
from lightcurver_amd.joint import StarPhotometryBatch
from lightcurver_amd.synthetic import make_roi_dataset
G, E, n, iters = [int(x) for x in sys.argv[1:5]]
base = make_roi_dataset(E=E, M=1, n=n, ss=2, seed=77, with_background=False)
ctx = _lib.Context(0)
stacks = [(base['data'], base['noisemap'].astype(np.float64) ** 2, base['psf'])] * G
t0 = time.perf_counter(); b = StarPhotometryBatch(stacks, 2, 1, ctx); ctx.synchronize(); print('create', time.perf_counter() - t0)
a = np.tile(np.asarray(base['truth']['a']) * 0.9, G)
b.set_params(a=a, c_x=np.zeros(G), c_y=np.zeros(G), dx=np.zeros(G * E), dy=np.zeros(G * E), alpha=np.zeros(G * E), mean=np.zeros(G * E))
b.set_loss(); b.set_free(['a', 'c_x', 'c_y', 'dx', 'dy'])
b.run_adabelief(5, init_learning_rate=1e-3); ctx.synchronize()
ctx.timer_start(); t0 = time.perf_counter()
b.run_adabelief(iters, init_learning_rate=1e-3)
ms = ctx.timer_stop(); wall = time.perf_counter() - t0
print(f'G={G} E={E} n={n}: {ms / iters * 1e3:.1f} us/iter (device), wall {wall / iters * 1e6:.1f} us/iter')
t0 = time.perf_counter(); h = b.loss_history(); print('history', time.perf_counter() - t0, h[0, 0], h[0, -1])
t0 = time.perf_counter(); m = b.model(); print('model', time.perf_counter() - t0)
t0 = time.perf_counter(); s = b.fisher_flux_sigma(); print('fisher', time.perf_counter() - t0)
