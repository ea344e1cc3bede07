"""Float32 end results of the PSF pixel-grid fit of tests/golden/psf_converged.npz, by two CPU implementations that share
no arithmetic with each other nor with the HIP kernels:

  * ``oracle/model.py`` + ``oracle/optim.py`` run in torch.float32 (FFT convolution, autograd) - the precision the
    reference itself computes in (cutouts are float32, lightcurver/processes/cutout_making.py:48-49; JAX default fp32);
  * ``oracle/psf_cpu.c`` built with real = float (direct separable sums, hand-derived adjoints).

Both run the same 3000 AdaBelief iterations from the same starting point as the float64 fixture.  Their distance from the
float64 end result is the spread fp32 implementations of this fit have; tests/test_north_star_gpu.py asserts that the
HIP path lies inside it (``psf_converged_f32.npz``).

PARITY UNPINNED (oracle/__init__.py): these bracket the oracle's fp32 behaviour, not STARRED's.
Run from the repository root (a few minutes of CPU):   python tests/golden/make_converged_f32_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as om, optim as oo, psf_cpu  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = np.load(os.path.join(HERE, 'psf_converged.npz'))
    ss, T = int(g['ss']), int(g['T'])
    F, S, n, _ = g['data'].shape
    N = n * ss
    J = om.n_scales(N)
    f32 = torch.float32
    out = dict(T=T)
    chi2_t, loss_t, a_t, x0_t, y0_t = [], [], [], [], []
    for f in range(F):
        data, sig2 = om.T(g['data'][f], f32), om.T(g['noisemap'][f], f32) ** 2
        mask = om.T(g['masks'][f].astype(np.float64), f32)
        mo, st = g['moffat'][f], g['stars0'][f]
        p = dict(fwhm_x=om.T(mo[0], f32), fwhm_y=om.T(mo[1], f32), phi=om.T(mo[2], f32), beta=om.T(mo[3], f32),
                 B=om.T(g['B0'][f], f32).reshape(-1), a=om.T(st[:, 0], f32), x0=om.T(st[:, 1], f32),
                 y0=om.T(st[:, 2], f32), sky=om.T(st[:, 3], f32))
        Wf = torch.cat([om.T(g['W'][f], f32), torch.zeros(1, N, N, dtype=f32)])
        fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=Wf, lam_scales=1.0, lam_hf=1.0)
        pf, lh, l0 = oo.adabelief(fn, p, ['B', 'a', 'x0', 'y0'], 1e-4, T, schedule=True)
        chi2_t.append(om.reduced_chi2(data.double(), om.psf_model(pf, ss, n).double(), sig2.double(), mask.double()))
        loss_t.append(lh[-1])
        a_t.append(pf['a'].numpy())
        x0_t.append(pf['x0'].numpy())
        y0_t.append(pf['y0'].numpy())
        print('torch fp32 frame', f, 'chi2', chi2_t[-1], 'rel. to f64', abs(chi2_t[-1] - g['chi2'][f]) / g['chi2'][f],
              'loss rel.', abs(lh[-1] - g['loss_final'][f]) / g['loss_final'][f])
    out.update(chi2_torch_f32=np.array(chi2_t), loss_torch_f32=np.array(loss_t), a_torch_f32=np.stack(a_t),
               x0_torch_f32=np.stack(x0_t), y0_torch_f32=np.stack(y0_t))

    # the C restatement, fp32
    Tm = np.stack([om.moffat(N, ss, *[om.T(v) for v in g['moffat'][f]]).numpy() for f in range(F)])
    w = (g['masks'] / g['noisemap'].astype(np.float64) ** 2).astype(np.float32)
    st = psf_cpu.PsfCpuState(g['data'], w, ss, Tm, g['W'][:, :J], g['B0'], g['stars0'])
    hist = st.run_adabelief(T, lr0=1e-4, schedule=True, threads=2)
    ev = st.evaluate(1.0, 1.0)
    nvalid = g['masks'].reshape(F, -1).sum(axis=1)
    chi2_c = ev['chi2'].astype(np.float64) / nvalid
    for f in range(F):
        print('C fp32 frame', f, 'chi2', chi2_c[f], 'rel. to f64', abs(chi2_c[f] - g['chi2'][f]) / g['chi2'][f],
              'loss rel.', abs(hist[f, -1] - g['loss_final'][f]) / g['loss_final'][f])
    out.update(chi2_c_f32=chi2_c, loss_c_f32=hist[:, -1].astype(np.float64), a_c_f32=st.stars[:, :, 0].copy(),
               x0_c_f32=st.stars[:, :, 1].copy(), y0_c_f32=st.stars[:, :, 2].copy())
    np.savez_compressed(os.path.join(HERE, 'psf_converged_f32.npz'), **out)


if __name__ == '__main__':
    main()
