"""oracle/joint_cpu.c (the C restatement of the joint fit WITH the pixelated background that bench.py times as the CPU
baseline of the joint-fit entries, kind "port") against oracle/model.py: two implementations that share no arithmetic
(radix-2 FFTs written in C, hand-derived adjoints - correlation, scatter of the bilinear weights, the adjoint recursion of
the a-trous cascade - there; torch.fft and autograd here).  The loop it stands for: lightcurver/processes/roi_modelling.py:308-334.
PARITY UNPINNED (oracle/__init__.py)."""
import numpy as np
import pytest
import torch

from lightcurver_amd.synthetic import make_roi_dataset
from oracle import model as om, optim as oo
from oracle.joint_cpu import JointCpu

FREE = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h']
LOSS = dict(lam_scales=1.0, lam_hf=1.5, lam_positivity=20.0, lam_positivity_ps=3.0, lam_pts_source=0.3, lam_flux_uniformity=0.7)


def _problem(E, M, n, seed, rotate):
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=seed)
    p = {k: np.asarray(v, np.float64) for k, v in ds['truth'].items()}
    rng = np.random.default_rng(seed)
    p['a'] = p['a'] * rng.uniform(0.8, 1.2, p['a'].shape)
    p['a'][0] = -0.05 * abs(p['a'][0])                       # a negative flux: the positivity term of the point sources is active
    p['c_x'] = p['c_x'] + rng.uniform(-0.2, 0.2, M)
    p['dx'] = p['dx'] + rng.uniform(-0.3, 0.3, E)
    p['dy'] = p['dy'] + rng.uniform(-0.3, 0.3, E)
    p['mean'] = rng.uniform(-1e-3, 1e-3, E)
    p['alpha'] = rng.uniform(-4.0, 4.0, E) if rotate else np.zeros(E)
    p['h'] = p['h'] + 2e-3 * rng.standard_normal(p['h'].shape)   # some negative pixels: the positivity term of h is active
    return ds, p


def _oracle_fn(ds, sig2, W):
    d, s2, ps = om.T(ds['data']), om.T(sig2), om.T(ds['psf'])
    Wt = None if W is None else om.T(W)
    return lambda q: om.deconv_loss(q, d, s2, ps, 2, W=Wt, lam_scales=LOSS['lam_scales'], lam_hf=LOSS['lam_hf'],
                                    lam_pos=LOSS['lam_positivity'], lam_pos_ps=LOSS['lam_positivity_ps'],
                                    lam_pts=LOSS['lam_pts_source'], lam_fu=LOSS['lam_flux_uniformity'])


@pytest.mark.parametrize('E,M,n,rotate,with_W', [(3, 2, 16, False, False), (2, 1, 16, True, True), (3, 3, 24, True, True)])
def test_loss_gradient_and_model_equal_the_oracle(E, M, n, rotate, with_W):
    ds, p = _problem(E, M, n, 50 + n + E, rotate)
    sig2 = ds['noisemap'].astype(np.float64) ** 2
    N = 2 * n
    W = None
    if with_W:
        W = om.propagate_noise_deconv(om.T(sig2), om.T(ds['psf']), 2).numpy()
    c = JointCpu(ds['data'], sig2, ds['psf'], 2, M, double=True, threads=2)
    c.set_params(**p)
    c.set_loss(W=W, **LOSS)
    loss, g, model = c.eval(threads=2, want_model=True)
    po = {k: om.T(v) for k, v in p.items()}
    Lo, go = oo.value_and_grad(_oracle_fn(ds, sig2, W), po, FREE)
    assert abs(loss - float(Lo)) / abs(float(Lo)) < 1e-11
    for k in FREE:
        gk = go[k].numpy()
        assert np.abs(g[k] - gk).max() / np.abs(gk).max() < 1e-9, k
    mo = om.deconv_model(po, om.T(ds['psf']), 2, n).numpy()
    assert np.abs(model - mo).max() / np.abs(mo).max() < 1e-12
    c.close()


def test_adabelief_trajectory_equals_the_oracle_and_does_not_depend_on_the_threads():
    E, M, n, T = 4, 2, 16, 30
    ds, p = _problem(E, M, n, 9, False)
    sig2 = ds['noisemap'].astype(np.float64) ** 2
    runs = []
    for thr in (1, 3):
        c = JointCpu(ds['data'], sig2, ds['psf'], 2, M, double=True, threads=thr)
        c.set_params(**p)
        c.set_loss(**LOSS)
        runs.append((c.run(T, lr0=1e-3, schedule=True, free=FREE, threads=thr), {k: v.copy() for k, v in c.p.items()}))
        c.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and all(np.array_equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1])
    po = {k: om.T(v) for k, v in p.items()}
    pf, lh, l0 = oo.adabelief(_oracle_fn(ds, sig2, None), po, FREE, 1e-3, T, schedule=True)
    hist, final = runs[0]
    assert abs(hist[0] - l0) / abs(l0) < 1e-11 and np.abs(hist[1:] - np.array(lh)).max() / abs(l0) < 1e-9
    for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean'):
        assert np.abs(final[k] - pf[k].numpy()).max() < 1e-9 * (1 + np.abs(pf[k].numpy()).max()), k
    dh = np.abs(final['h'] - pf['h'].numpy())
    assert np.median(dh) < 1e-10 and dh.max() < 2.5e-3 * T   # a sign flip of a ~0 starlet coefficient moves a pixel by <= 2 lr per step


def test_float32_build_tracks_the_float64_one():
    """The float32 build is what bench.py times: same problem, 1e-4 of the loss after 10 iterations."""
    E, M, n = 3, 2, 16
    ds, p = _problem(E, M, n, 11, False)
    sig2 = ds['noisemap'].astype(np.float64) ** 2
    out = []
    for dbl in (True, False):
        c = JointCpu(ds['data'], sig2, ds['psf'], 2, M, double=dbl, threads=2)
        c.set_params(**p)
        c.set_loss(**LOSS)
        out.append(c.run(10, lr0=1e-4, free=FREE, threads=2))
        c.close()
    assert np.all(np.isfinite(out[1])) and np.abs(out[1] - out[0]).max() / abs(out[0][0]) < 1e-4
