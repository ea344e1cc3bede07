import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd import _lib
if os.environ.get('LCMI_DBG_LIB'): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['LCMI_DBG_LIB'])
from tests.test_psf_gpu import _setup
ctx = _lib.Context(0)
n, S, F, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
out = []
for single in (False, True):
    if single: os.environ['LCMI_PSF_SINGLE_WG'] = '1'
    ds, plist, b = _setup(n, 2, F, S, 900 + n, ctx, jitter=0.1)
    b.propagate_noise(); b.set_regularization(None, 1.0, 1.0)
    b.run_adabelief(T, init_learning_rate=1e-4, schedule_learning_rate=True)
    out.append((b.loss_history(), b.get_grid(), b.get_stars()))
h0, h1 = out[0][0], out[1][0]
print('hist equal', np.array_equal(h0, h1), 'first diff iter per frame', [int(np.argmax(h0[f] != h1[f])) if (h0[f] != h1[f]).any() else -1 for f in range(F)])
g0, g1 = out[0][1].reshape(F, 2 * n, 2 * n), out[1][1].reshape(F, 2 * n, 2 * n)
d = g0 != g1
print('grid differing px', d.sum(), 'of', d.size, 'per frame', d.reshape(F, -1).sum(1)[:10])
if d.any():
    f = np.argwhere(d.reshape(F, -1).sum(1) > 0)[0, 0]
    rows = np.argwhere(d[f].sum(1) > 0).ravel(); cols = np.argwhere(d[f].sum(0) > 0).ravel()
    print('frame', f, 'rows', rows[:40], 'cols', cols[:70])
    print('maxabs', np.abs(g0 - g1).max())
