"""Diagnostic: phase shares of the joint epoch kernel (liblcmi_dbg.so, -DLC_STAMPS)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightcurver_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'liblcmi_dbg.so')
from lightcurver_amd.joint import JointFit
from lightcurver_amd.synthetic import make_roi_dataset
E, n, M = [int(x) for x in sys.argv[1:4]]
with_h = len(sys.argv) < 5 or sys.argv[4] != 'noh'
ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
ctx = _lib.Context(0)
j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
p = dict(ds['truth'])
if not with_h: p['h'] = np.zeros_like(p['h'])
j.set_params(**p)
j.set_loss(lam_scales=1.0, lam_hf=1.0)
j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean'] + (['h'] if with_h else []))
j.run_adabelief(20, init_learning_rate=1e-4); ctx.synchronize()
out = (C.c_longlong * 32)()
_lib.lib().lc_debug_get_jstamps.argtypes = [C.POINTER(C.c_longlong)]
assert _lib.lib().lc_debug_get_jstamps(out) == 0
s = np.array(out[:], dtype=np.int64)
names = ['tables', 'A scene+row FFT', 'B columns', 'C inv rows+resid+fwd rows', "B' adjoint columns", "C' inv rows + grads", 'reductions', "D T^T gather"]
tot = s[8] - s[0]
for k, nm in enumerate(names):
    print(f'{nm:28s} {s[k+1]-s[k]:8d} cycles {100*(s[k+1]-s[k])/tot:5.1f}%')
print('total', tot)
print('phase B last sweep of wave 0: loads', s[10]-s[9], 'fwd fft', s[11]-s[10], 'times spectrum', s[12]-s[11], 'inv fft', s[13]-s[12], 'store', s[14]-s[13])
if j.cluster_info()[0]:
    # cluster form: stamps 16 + k sit at the entry of sync k (k = 0: end of the prologue), stamps 1 .. 6 behind the syncs
    print('cluster form,', j.cluster_info()[0], 'workgroups per epoch, all on one XCD:', bool(s[30]), '; workgroup 0: compute / sync per phase (cycles)')
    after = [s[1], s[2], s[3], s[4], s[5], None]   # stamp taken right after sync k (k = 0 .. 4); sync 5 -> D
    prev = s[0]
    for k, nm in enumerate(['prologue', 'A', 'B', 'C', "B'", "C'"]):
        print(f'  {nm:10s} compute {s[16 + k] - prev:8d}   sync {(after[k] - s[16 + k]) if after[k] is not None else -1:8d}')
        prev = after[k] if after[k] is not None else s[16 + k]
    print(f'  totals + D from the entry of the last sync: {s[8] - s[21]:8d}')
