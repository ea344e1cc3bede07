"""Soak of the device loop's synchronisation (gate kernel, epoch-carried wait, cluster form, in-kernel flag waits): many fits of
many sizes back to back in one process, every run must end without a time-out report and with a decreasing, finite loss.
python tools/joint_soak.py [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightcurver_amd import _lib
from lightcurver_amd.joint import JointFit
from lightcurver_amd.synthetic import make_roi_dataset
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = _lib.Context(0)
cases = [(8, 64, 2, 1500), (25, 64, 2, 2000), (32, 64, 2, 1500), (33, 64, 2, 1000), (40, 64, 2, 1000), (100, 64, 2, 1000), (200, 64, 2, 1000),
         (4, 128, 4, 300), (32, 128, 4, 300), (64, 128, 4, 300), (125, 128, 4, 300), (5, 32, 2, 1500), (6, 24, 1, 1500)]
bad = 0
for r in range(rep):
    for E, n, M, T in cases:
        ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=100 + r)
        j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
        p = dict(ds['truth']); p['a'] = np.asarray(p['a']) * 0.9
        j.set_params(**p)
        W = j.propagate_noise()
        j.set_loss(W=W, lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
        j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
        t0 = time.time()
        try:
            for _ in range(3):
                j.run_adabelief(T // 3, init_learning_rate=1e-4, schedule_learning_rate=False)
            h = np.asarray(j.loss_history())
            ok = np.all(np.isfinite(h)) and h[-1] < h[0]
            print(f'rep {r} E={E} n={n} M={M}: {(time.time() - t0) / T * 1e6:.1f} us/iter, cluster {j.cluster_info()}, loss {h[0]:.1f} -> {h[-1]:.1f} {"ok" if ok else "BAD"}', flush=True)
            bad += 0 if ok else 1
        except Exception as e:
            print(f'rep {r} E={E} n={n} M={M}: FAILED {e}', flush=True)
            bad += 1
        j.close()
print('soak', 'FAILED' if bad else 'clean', bad)
sys.exit(1 if bad else 0)
