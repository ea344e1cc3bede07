"""Diagnostic: where the fused reduction + update launch spends its time (liblcmi_dbg.so, -DLC_STAMPS; 100 MHz wall clock):
python tools/update_stamps.py E n M"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightcurver_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'liblcmi_dbg.so')
from lightcurver_amd.joint import JointFit
from lightcurver_amd.synthetic import make_roi_dataset
E, n, M = [int(x) for x in sys.argv[1:4]]
ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
ctx = _lib.Context(0)
j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
j.set_params(**ds['truth'])
W = j.propagate_noise()
j.set_loss(W=W, lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
j.run_adabelief(50, init_learning_rate=1e-4); ctx.synchronize()
out = (C.c_longlong * 16)()
_lib.lib().lc_debug_get_ustamps.argtypes = [C.POINTER(C.c_longlong)]
assert _lib.lib().lc_debug_get_ustamps(out) == 0
s = np.array(out[:], dtype=np.int64)
t0 = min(s[0], s[2])
names = {0: 'image block 0 starts', 1: 'last image block done', 2: 'scalar block starts', 3: 'scalars reduced', 4: 'flag seen',
         5: 'fluxes stepped', 6: 'positions stepped', 7: 'loss partials combined', 8: 'scalar block done'}
for k in sorted(names, key=lambda k: s[k]):
    print(f'{(s[k] - t0) * 10:8d} ns  {names[k]}')
