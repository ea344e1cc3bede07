"""Cross-check of the two regulariser implementations inside the device loop: matrix-core form (default, N >= 128) against
the a-trous cascade kernels (LCMI_REG_CASCADE=1).  python tools/reg_ab.py E n M iters"""
import os, subprocess, sys
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lightcurver_amd import _lib
    from lightcurver_amd.joint import JointFit
    from lightcurver_amd.synthetic import make_roi_dataset
    E, n, M, iters = [int(x) for x in sys.argv[2:6]]
    out = sys.argv[6]
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
    ctx = _lib.Context(0)
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
    p = dict(ds['truth']); p['a'] = p['a'] * 0.9
    j.set_params(**p)
    W = j.propagate_noise()
    j.set_loss(W=W, lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
    j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
    j.run_adabelief(5, init_learning_rate=1e-4, schedule_learning_rate=False); ctx.synchronize()
    ctx.timer_start()
    j.run_adabelief(iters, init_learning_rate=1e-4, schedule_learning_rate=False)
    ms = ctx.timer_stop()
    got = j.get_params()
    np.savez(out, hist=j.loss_history(), us=ms / iters * 1e3, **{k: np.asarray(v) for k, v in got.items()})
    sys.exit(0)
args = sys.argv[1:5]
res = {}
for name, env in (('mfma', {}), ('cascade', {'LCMI_REG_CASCADE': '1'})):
    out = f'/tmp/reg_ab_{name}.npz'
    subprocess.run([sys.executable, __file__, '--child', *args, out], check=True, env={**os.environ, **env})
    res[name] = np.load(out)
a, b = res['mfma'], res['cascade']
print('us/iter  mfma %.1f  cascade %.1f' % (float(a['us']), float(b['us'])))
print('loss history max rel diff', np.max(np.abs(a['hist'] - b['hist']) / np.abs(b['hist'])), 'first', a['hist'][0], b['hist'][0], 'last', a['hist'][-1], b['hist'][-1])
for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'):
    d = np.max(np.abs(a[k] - b[k])); s = np.max(np.abs(b[k])) + 1e-30
    print(f'{k:5s} max abs diff {d:.3e}  (scale {s:.3e})')
