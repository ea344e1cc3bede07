"""Diagnostic: where the row-block regulariser kernel (csrc/joint_reg_rows.h) spends its cycles (liblcmi_dbg.so, -DLC_STAMPS;
workgroup 0 = row block 0, scales 1 .. 4)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightcurver_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'liblcmi_dbg.so')
from lightcurver_amd.joint import JointFit
from lightcurver_amd.synthetic import make_roi_dataset
E, n, M = 25, 64, 2
ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
ctx = _lib.Context(0)
j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
j.set_params(**dict(ds['truth']))
j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0)
j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
j.run_adabelief(20, init_learning_rate=1e-4); ctx.synchronize()
out = (C.c_longlong * 16)()
_lib.lib().lc_debug_get_rstamps.argtypes = [C.POINTER(C.c_longlong)]
assert _lib.lib().lc_debug_get_rstamps(out) == 0
s = np.array(out[:], dtype=np.int64)
for k, nm in enumerate(['X into LDS', 'forward products (T1, c)', 'S rows', 'U products', 'Z products', 'partial plane out + values']):
    print(f'{nm:28s} {s[k + 1] - s[k]:8d} cycles')
print('total', s[6] - s[0])
