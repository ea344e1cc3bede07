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
out = (C.c_longlong * 128)()
_lib.lib().lc_debug_get_stamps.argtypes = [C.POINTER(C.c_longlong)]
assert _lib.lib().lc_debug_get_stamps(out) == 0
allst = np.array(out[:], dtype=np.int64)
names = {0: 'start', 1: 'P1 + taps (to first barrier)', 40: 'conv: wave tasks + P5', 42: 'starlet', 44: 'l1 reduce',
         45: 'publish (stores, drain, barrier)', 46: 'flag + wait for partner', 47: 'read partner slab', 43: 'loss + update'}
order = [0, 1, 40, 42, 44, 45, 46, 47, 43]
for blk, label in ((0, 'block 0 (role 0 / single form)'), (64, 'block 8 (role 1)')):
    s = allst[blk:blk + 64]
    if s[43] == 0:
        continue
    print(label)
    keys = [k for k in order if s[k] != 0]
    tot = s[43] - s[0]
    prev = s[0]
    for k in keys[1:]:
        print(f'  {names[k]:34s} {s[k]-prev:8d} ticks  {100*(s[k]-prev)/tot:5.1f}%')
        prev = s[k]
    fine = {10: 'task start (prefetch issued)', 11: 'row pass', 12: 'column pass + reductions', 13: 'transposed column pass', 14: 'wave 0 done with its tasks', 15: 'barrier (other waves)', 16: 'per-star sums', 17: 'P5 transposed row pass'}
    if s[10]:
        print('   last group, wave 0, last task:')
        for a in range(11, 18):
            print(f'     {fine[a]:34s} {s[a]-s[a-1]:8d} ticks')
    if blk == 0 and s[20]:
        print('   waves done with their tasks, ticks after the first barrier:', [int(s[20 + w] - s[1]) for w in range(16) if s[20 + w]])
    print('  total ticks', tot, '(s_memtime ticks @100MHz => us:', tot / 100.0, ')')
if len(sys.argv) > 2 and sys.argv[2] == 'groups':   # non-WC kernels (N = 128): the four passes of every group of stars
    s = allst[:64]
    print('group: P2 row pass | P3 column pass + chi2 | P4 transposed column | P5 transposed row   (ticks, block 0, last iteration)')
    for g in range(8):
        a = [s[1 + 5 * g], s[2 + 5 * g], s[3 + 5 * g], s[4 + 5 * g]]
        nxt = s[1 + 5 * (g + 1)] if g < 7 and s[1 + 5 * (g + 1)] else s[40]
        if a[0]:
            print(f'  {g}: {a[1]-a[0]:7d} {a[2]-a[1]:7d} {a[3]-a[2]:7d} {nxt-a[3]:7d}')
