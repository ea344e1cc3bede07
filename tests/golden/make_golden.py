"""Generates the golden vectors under tests/golden/ from the float64 oracle on seeded synthetic inputs.

PARITY UNPINNED: the reference's arithmetic (STARRED) cannot be run or imported here and its own tests
hold no numeric fixture for this path, so these vectors pin the *oracle* (regression
anchor for oracle/ and target for the HIP path), not STARRED.  Run from the repository root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as om, optim as oo  # noqa: E402
from lightcurver_amd.synthetic import make_psf_dataset, make_roi_dataset  # noqa: E402
from tests import helpers as H  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def psf_case():
    ss, n, S = 2, 16, 3
    ds = make_psf_dataset(F=1, S=S, n=n, ss=ss, seed=2024)
    rng = np.random.default_rng(7)
    p = H.psf_initial_params(ds, 0, ss, rng, 0.2)
    data, sig2, mask = H.psf_oracle_inputs(ds, 0, ss)
    W = om.propagate_noise_psf(p, sig2, mask, ss)
    fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=W, lam_scales=1.0, lam_hf=1.0)
    free = ['fwhm_x', 'fwhm_y', 'phi', 'beta', 'a', 'x0', 'y0', 'sky', 'B']
    L, g = oo.value_and_grad(fn, p, free)
    pf, lh, l0 = oo.adabelief(fn, p, ['B', 'a', 'x0', 'y0'], 1e-4, 8, schedule=True)
    narrow, full = om.psf_outputs(pf, ss, n)
    out = dict(data=ds['data'][0], noisemap=ds['noisemap'][0], masks=ds['masks'][0], ss=ss,
               W=W.numpy(), loss=L, model=om.psf_model(p, ss, n).numpy(), traj_loss=np.array([l0] + lh),
               traj_B=pf['B'].numpy(), traj_a=pf['a'].numpy(), narrow=narrow.numpy(), full=full.numpy())
    for k, v in p.items():
        out['p_' + k] = v.numpy()
    for k in free:
        out['g_' + k] = g[k].numpy()
    np.savez_compressed(os.path.join(HERE, 'psf_small.npz'), **out)


def joint_case():
    E, M, n, ss = 3, 2, 16, 2
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=2025, alpha_sigma=1.0)
    rng = np.random.default_rng(11)
    t = ds['truth']
    p = {k: np.array(v, dtype=np.float64) for k, v in t.items()}
    p['a'] *= rng.uniform(0.9, 1.1, p['a'].shape)
    p['c_x'] += rng.normal(0, 0.1, M)
    p['dx'] += rng.normal(0, 0.05, E)
    p['mean'] = rng.normal(0, 1e-3, E)
    p['h'] = p['h'] * rng.uniform(0.8, 1.2, p['h'].shape) + 2e-3 * rng.standard_normal(p['h'].shape)
    po = {k: om.T(v) for k, v in p.items()}
    data, sig2, psf = om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])
    W = om.propagate_noise_deconv(sig2, psf, ss)
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, W=W, lam_scales=1.0, lam_hf=1.0, lam_pos=10.0, lam_fu=0.5)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean']
    L, g = oo.value_and_grad(fn, po, free)
    pf, lh, l0 = oo.adabelief(fn, po, free, 1e-3, 8, schedule=True)
    out = dict(data=ds['data'], noisemap=ds['noisemap'], psf=ds['psf'], ss=ss, W=W.numpy(), loss=L,
               model=om.deconv_model(po, psf, ss, n).numpy(), traj_loss=np.array([l0] + lh),
               traj_a=pf['a'].numpy(), fisher=om.fisher_flux_sigma(po, sig2, psf, ss).numpy())
    for k, v in p.items():
        out['p_' + k] = v
    for k in free:
        out['g_' + k] = g[k].numpy()
    np.savez_compressed(os.path.join(HERE, 'joint_small.npz'), **out)


if __name__ == '__main__':
    torch.set_num_threads(1)
    psf_case()
    joint_case()
    print('golden vectors written to', HERE)
