"""Diagnostic: phase shares of one PSF-fit iteration from in-kernel clock stamps (liblcmi_dbg.so,
built with -DLC_STAMPS).  Never used for timing claims."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightcurver_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'liblcmi_dbg.so')
from lightcurver_amd.psf_batch import PsfBatch
from lightcurver_amd.synthetic import make_psf_dataset
cfg = dict(F=100, S=8, n=int(sys.argv[1]) if len(sys.argv) > 1 else 32, ss=2, seed=1)
ds = make_psf_dataset(**cfg)
ctx = _lib.Context(0)
w = (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)
b = PsfBatch(ds['data'], w, 2, ctx)
F, S = cfg['F'], cfg['S']
g = ds['fwhm_guess']; f0 = np.sqrt(np.maximum(g * g - 1, 1.0))
b.set_moffat(np.stack([f0, f0, np.zeros(F), np.full(F, 2.5)], -1))
st = np.zeros((F, S, 4), np.float32); st[..., 0] = (ds['data'] * ds['masks']).sum((-1, -2)); b.set_stars(st)
b.set_grid(None); b.propagate_noise(); b.set_regularization(None, 1.0, 1.0)
b.run_adabelief(50, init_learning_rate=1e-4); ctx.synchronize()
out = (C.c_longlong * 64)()
_lib.lib().lc_debug_get_stamps.argtypes = [C.POINTER(C.c_longlong)]
assert _lib.lib().lc_debug_get_stamps(out) == 0
s = np.array(out[:], dtype=np.int64)
names = {0: 'start', 1: 'P1 + taps (to first barrier)', 40: 'conv: wave tasks + P5', 42: 'starlet', 43: 'loss+update'}
names_old = {0: 'start', 1: 'g0 taps done', 2: 'g0 P2 row', 3: 'g0 P3 col+res', 4: 'g0 P4 colT', 6: 'g1 taps (incl g0 P5)',
         7: 'g1 P2', 8: 'g1 P3', 9: 'g1 P4', 40: 'groups end (g1 P5)', 41: 'starlet fwd', 42: 'starlet bwd', 43: 'loss+update'}
keys = [k for k in sorted(names) if s[k] != 0]
tot = s[43] - s[0]
prev = s[0]
for k in keys[1:]:
    print(f'{names[k]:28s} {s[k]-prev:8d} ticks  {100*(s[k]-prev)/tot:5.1f}%')
    prev = s[k]
print('total ticks', tot, '(s_memtime ticks @100MHz => us:', tot / 100.0, ')')
