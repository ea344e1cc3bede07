"""oracle/joint_ps_cpu.c (the C restatement of the point-source-only joint fit that bench.py times as the star-photometry
CPU baseline) against oracle/model.py: two implementations that share no arithmetic (direct separable sums with derivative
taps there; full-frame Gaussian raster, FFT convolution and autograd here).  PARITY UNPINNED (oracle/__init__.py)."""
import numpy as np
import pytest
import torch

from lightcurver_amd.synthetic import make_roi_dataset
from oracle import model as om, optim as oo
from oracle.joint_ps_cpu import JointPsCpu


def _problem(E, M, n, seed):
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=seed, with_background=False)
    p = {k: np.asarray(v, np.float64) for k, v in ds['truth'].items()}
    rng = np.random.default_rng(seed)
    p['a'] = p['a'] * rng.uniform(0.8, 1.2, p['a'].shape)
    p['c_x'] = p['c_x'] + rng.uniform(-0.2, 0.2, M)
    p['dx'] = p['dx'] + rng.uniform(-0.2, 0.2, E)
    p['dy'] = p['dy'] + rng.uniform(-0.2, 0.2, E)
    p['mean'] = rng.uniform(-1e-3, 1e-3, E)
    p['alpha'] = np.zeros(E)
    p['h'] = np.zeros((n * 2) ** 2)
    return ds, p


@pytest.mark.parametrize('E,M,n', [(3, 1, 16), (4, 2, 24), (2, 3, 32)])
def test_loss_gradient_and_model_equal_the_oracle(E, M, n):
    ds, p = _problem(E, M, n, 40 + n)
    sig2 = ds['noisemap'].astype(np.float64) ** 2
    c = JointPsCpu(ds['data'], sig2, ds['psf'], 2, M, double=True)
    c.set_params(**{k: p[k] for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean')})
    loss, g, model = c.eval(threads=2, want_model=True)
    po = {k: om.T(v) for k, v in p.items()}
    fn = lambda q: om.deconv_loss(q, om.T(ds['data']), om.T(sig2), om.T(ds['psf']), 2)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean']
    Lo, go = oo.value_and_grad(fn, po, free)
    assert abs(loss - float(Lo)) / float(Lo) < 1e-11
    for k, gk in zip(free, go):
        gk = gk.numpy() if torch.is_tensor(gk) else np.asarray(go[k])
        assert np.abs(g[k] - gk).max() / np.abs(gk).max() < 1e-9, k
    mo = om.deconv_model(po, om.T(ds['psf']), 2, n).numpy()
    assert np.abs(model - mo).max() / np.abs(mo).max() < 1e-12


def test_adabelief_trajectory_equals_the_oracle_and_does_not_depend_on_the_threads():
    E, M, n, T = 5, 1, 16, 40
    ds, p = _problem(E, M, n, 7)
    sig2 = ds['noisemap'].astype(np.float64) ** 2
    runs = []
    for thr in (1, 3):
        c = JointPsCpu(ds['data'], sig2, ds['psf'], 2, M, double=True)
        c.set_params(**{k: p[k] for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean')})
        runs.append((c.run(T, lr0=1e-3, schedule=True, threads=thr), {k: v.copy() for k, v in c.p.items()}))
    assert np.array_equal(runs[0][0], runs[1][0]) and all(np.array_equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1])
    po = {k: om.T(v) for k, v in p.items()}
    fn = lambda q: om.deconv_loss(q, om.T(ds['data']), om.T(sig2), om.T(ds['psf']), 2)
    pf, lh, l0 = oo.adabelief(fn, po, ['a', 'c_x', 'c_y', 'dx', 'dy'], 1e-3, T, schedule=True)
    hist, final = runs[0]
    assert abs(hist[0] - l0) / l0 < 1e-11 and np.abs(hist[1:] - np.array(lh)).max() / l0 < 1e-9
    for k in ('a', 'c_x', 'c_y', 'dx', 'dy'):
        assert np.abs(final[k] - pf[k].numpy()).max() < 1e-8 * max(1.0, np.abs(final[k]).max()), k


def test_fp32_build_follows_the_fp64_one():
    E, M, n = 6, 1, 32
    ds, p = _problem(E, M, n, 9)
    sig2 = ds['noisemap'].astype(np.float64) ** 2
    out = []
    for dbl in (True, False):
        c = JointPsCpu(ds['data'], sig2, ds['psf'], 2, M, double=dbl)
        c.set_params(**{k: p[k] for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean')})
        out.append((c.run(30, threads=2), c.p['a'].astype(np.float64)))
    assert np.abs(out[0][0] - out[1][0]).max() / out[0][0][0] < 1e-4
    assert np.abs(out[0][1] - out[1][1]).max() / np.abs(out[0][1]).max() < 1e-4
