"""Pins of the CPU oracle (no GPU): library known-answer checks for every primitive, derivative checks,
and the committed golden vectors.  The oracle is a restatement of STARRED with PARITY UNPINNED
(DESIGN.md section 2): these tests pin it against scipy and against itself, not against STARRED."""
import math
import os

import numpy as np
import pytest
import scipy.ndimage
import scipy.signal
import torch

from oracle import model as om, optim as oo
from lightcurver_amd.synthetic import make_psf_dataset, make_roi_dataset
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_conv_same_is_scipy_fftconvolve():
    rng = np.random.default_rng(0)
    for N in (15, 16, 24):
        a, k = rng.standard_normal((2, N, N)), rng.standard_normal((2, N, N))
        ref = np.stack([scipy.signal.fftconvolve(a[i], k[i], mode='same') for i in range(2)])
        assert np.abs(om.conv_same(om.T(a), om.T(k)).numpy() - ref).max() < 1e-12


def test_bilinear_is_map_coordinates_nearest():
    rng = np.random.default_rng(1)
    h = rng.standard_normal((16, 16))
    Ys, Xs = rng.uniform(-3, 19, (2, 16, 16))
    ref = scipy.ndimage.map_coordinates(h, [Ys, Xs], order=1, mode='nearest')
    assert np.abs(om.bilinear_clamp(om.T(h), om.T(Ys), om.T(Xs)).numpy() - ref).max() < 1e-13


@pytest.mark.parametrize('N', [16, 32, 48])
def test_starlet_reconstruction_norms_and_adjoint(N):
    rng = np.random.default_rng(N)
    J = om.n_scales(N)
    img = om.T(rng.standard_normal((N, N)))
    st = om.starlet(img, J)
    assert st.shape == (J + 1, N, N)
    assert (st.sum(0) - img).abs().max() < 1e-13  # exact reconstruction
    norms = om.starlet_norms(64, 6).numpy()
    assert np.allclose(norms[:3], [0.8908, 0.2007, 0.0855], atol=2e-4)  # published starlet scale norms
    # adjoint by autograd satisfies the dot-product test
    x = om.T(rng.standard_normal((N, N))).requires_grad_(True)
    y = om.T(rng.standard_normal((J + 1, N, N)))
    (g,) = torch.autograd.grad((om.starlet(x, J) * y).sum(), x)
    z = om.T(rng.standard_normal((N, N)))
    assert abs(float((om.starlet(z, J) * y).sum()) - float((g * z).sum())) < 1e-10


def test_blocksum_and_upsample_are_adjoint():
    rng = np.random.default_rng(2)
    a, b = om.T(rng.standard_normal((3, 8, 8))), om.T(rng.standard_normal((3, 4, 4)))
    assert abs(float((om.blocksum(a, 2) * b).sum()) - float((a * om.upsample_rep(b, 2)).sum())) < 1e-12


def test_gaussian_is_unit_flux_and_moffat_normalised():
    g = om.gaussian_stack(32, om.T([15.3]), om.T([16.8]), om.T([1.0]))
    assert abs(float(g.sum()) - 1.0) < 1e-5
    m = om.moffat(32, 2, om.T(3.0), om.T(2.5), om.T(0.4), om.T(3.0))
    assert abs(float(m.sum()) - 1.0) < 1e-14 and int(torch.argmax(m)) == 15 * 32 + 15  # peak on the zero-lag index


def _small_joint(seed=3, alpha=1.0):
    ds = make_roi_dataset(E=2, M=2, n=8, ss=2, seed=seed, alpha_sigma=alpha)
    rng = np.random.default_rng(seed)
    p = {k: om.T(v) for k, v in ds['truth'].items()}
    p['h'] = p['h'] + om.T(1e-3 * rng.standard_normal(p['h'].shape))
    p['mean'] = om.T(1e-3 * rng.standard_normal(2))
    # keep the shifts off the integers: order-1 interpolation has a kink where a sample hits a pixel centre
    p['dx'] = p['dx'] + om.T([0.137, -0.211])
    p['dy'] = p['dy'] + om.T([-0.173, 0.119])
    return ds, p, om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])


def test_deconv_gradient_matches_finite_differences():
    ds, p, data, sig2, psf = _small_joint()
    W = om.propagate_noise_deconv(sig2, psf, 2)
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, 2, W=W, lam_scales=1.0, lam_hf=0.5, lam_pos=3.0, lam_fu=0.3)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean']
    L, g = oo.value_and_grad(fn, p, free)
    eps = 1e-6
    for k in free:
        for idx in range(p[k].numel()):
            q = {kk: v.clone() for kk, v in p.items()}
            q[k].view(-1)[idx] += eps
            lp = float(fn(q))
            q[k].view(-1)[idx] -= 2 * eps
            lm = float(fn(q))
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - float(g[k].view(-1)[idx])) < 2e-5 * max(1.0, abs(fd)), (k, idx)


def test_psf_model_centre_convention_and_flux():
    """A star with x0 = y0 = 0 and B = 0 is centred on the stamp and a_i is its total flux."""
    n, ss = 16, 2
    p = dict(fwhm_x=om.T(3.0), fwhm_y=om.T(3.0), phi=om.T(0.0), beta=om.T(3.0), B=torch.zeros(32 * 32, dtype=om.DT),
             a=om.T([2.5]), x0=om.T([0.0]), y0=om.T([0.0]), sky=om.T([0.0]))
    m = om.psf_model(p, ss, n)[0]
    yy, xx = np.mgrid[0:n, 0:n]
    cx = float((m * om.T(xx)).sum() / m.sum())
    cy = float((m * om.T(yy)).sum() / m.sum())
    # (a wrong zero-lag convention would show as a 0.25 px offset; the 2e-3 residual is the Moffat wing truncation)
    assert abs(cx - 7.5) < 1e-2 and abs(cy - 7.5) < 1e-2
    assert abs(float(m.sum()) - 2.5) < 2e-2  # Moffat wings leave the stamp


def test_adabelief_first_step_closed_form():
    p = {'x': om.T([1.0, -2.0])}
    fn = lambda q: (q['x'] ** 2).sum()
    pf, lh, l0 = oo.adabelief(fn, p, ['x'], 0.1, 1, schedule=False)
    # m = 0.1 g, s = 0.001 (0.9 g)^2 + eps_root -> update = g / (0.9 |g|) -> step = lr / 0.9 * sign(g)
    assert np.allclose(pf['x'].numpy(), [1.0 - 0.1 / 0.9, -2.0 + 0.1 / 0.9], atol=1e-9)
    assert l0 == 5.0 and len(lh) == 1
    assert abs(oo.learning_rate(20, 1e-3, True) - 1e-3 * 0.99 ** 2) < 1e-15


def test_noise_propagation_matches_monte_carlo():
    """W_j is the standard deviation of the chi2-gradient noise per starlet scale (SLIT == E[MC]) away from
    the image borders, where the shift-invariant formula ignores the edge replication of the starlet."""
    ds = make_roi_dataset(E=2, M=1, n=16, ss=2, seed=5)
    data, sig2, psf = om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])
    W = om.propagate_noise_deconv(sig2, psf, 2).numpy()
    rng = np.random.default_rng(0)
    E, n, _ = data.shape
    N, J = 2 * n, om.n_scales(2 * n)
    acc = np.zeros((J + 1, N, N))
    x0 = torch.zeros(E, N, N, dtype=om.DT, requires_grad=True)
    nsamp = 300
    for _ in range(nsamp):
        noise = om.T(rng.standard_normal((E, n, n))) / torch.sqrt(sig2)  # Sigma^-1 n, n ~ N(0, sigma^2)
        (g,) = torch.autograd.grad((om.blocksum(om.conv_same(x0, psf), 2) * noise).sum(), x0)
        acc += om.starlet(g.sum(0), J).numpy() ** 2
    mc = np.sqrt(acc / nsamp)
    for j in range(4):  # scales whose atom fits well inside the 32-pixel grid
        ratio = mc[j, 10:22, 10:22] / W[j, 10:22, 10:22]
        assert abs(ratio.mean() - 1) < 0.1, (j, ratio.mean())


def test_fisher_is_hessian_diagonal():
    ds, p, data, sig2, psf = _small_joint(seed=6)
    s = om.fisher_flux_sigma(p, sig2, psf, 2).numpy()
    fn = lambda a: 0.5 * (((data - om.deconv_model({**p, 'a': a}, psf, 2, 8)) ** 2) / sig2).sum()
    Hm = torch.autograd.functional.hessian(fn, p['a'])
    assert np.allclose(s, 1 / np.sqrt(np.diag(Hm.numpy())), rtol=1e-9)


def test_golden_psf_vectors():
    g = np.load(os.path.join(GOLD, 'psf_small.npz'))
    ss, n = int(g['ss']), g['data'].shape[-1]
    data, sig2, mask = om.T(g['data']), om.T(g['noisemap']) ** 2, om.T(g['masks'].astype(np.float64))
    p = {k[2:]: om.T(g[k]) for k in g.files if k.startswith('p_')}
    W = om.propagate_noise_psf(p, sig2, mask, ss)
    assert np.allclose(W.numpy(), g['W'], rtol=1e-10, atol=1e-12)
    fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=W, lam_scales=1.0, lam_hf=1.0)
    L, gr = oo.value_and_grad(fn, p, ['B', 'a', 'x0', 'beta'])
    assert abs(L - float(g['loss'])) < 1e-9 * abs(L)
    assert np.allclose(gr['B'].numpy(), g['g_B'], rtol=1e-8, atol=1e-10)
    assert np.allclose(gr['beta'].numpy(), g['g_beta'], rtol=1e-8)
    assert np.allclose(om.psf_model(p, ss, n).numpy(), g['model'], rtol=1e-10, atol=1e-13)


def test_golden_joint_vectors():
    g = np.load(os.path.join(GOLD, 'joint_small.npz'))
    ss, n = int(g['ss']), g['data'].shape[-1]
    data, sig2, psf = om.T(g['data']), om.T(g['noisemap']) ** 2, om.T(g['psf'])
    p = {k[2:]: om.T(g[k]) for k in g.files if k.startswith('p_')}
    W = om.propagate_noise_deconv(sig2, psf, ss)
    assert np.allclose(W.numpy(), g['W'], rtol=1e-9, atol=1e-12)
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, W=W, lam_scales=1.0, lam_hf=1.0, lam_pos=10.0, lam_fu=0.5)
    L, gr = oo.value_and_grad(fn, p, ['a', 'h', 'dx'])
    assert abs(L - float(g['loss'])) < 1e-9 * abs(L)
    assert np.allclose(gr['h'].numpy(), g['g_h'], rtol=1e-7, atol=1e-9)
    assert np.allclose(om.deconv_model(p, psf, ss, n).numpy(), g['model'], rtol=1e-10, atol=1e-13)
    pf, lh, l0 = oo.adabelief(fn, p, ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean'], 1e-3, 8, schedule=True)
    assert np.allclose(np.array([l0] + lh), g['traj_loss'], rtol=1e-9)


def test_synthetic_datasets_are_deterministic():
    a = make_psf_dataset(F=2, S=2, n=16, ss=2, seed=1)
    b = make_psf_dataset(F=2, S=2, n=16, ss=2, seed=1)
    assert np.array_equal(a['data'], b['data']) and a['data'].dtype == np.float32
    r = make_roi_dataset(E=3, M=2, n=16, ss=2, seed=1)
    assert r['data'].shape == (3, 16, 16) and r['psf'].shape == (3, 32, 32)
    assert np.allclose(r['psf'].sum((-1, -2)), 1.0, atol=1e-5)


def test_prep_oracle_follows_reference_lines():
    """oracle/prep.py against hand-evaluated values of the cited reference lines."""
    from oracle import prep as op
    data = np.array([[[4.0, -1.0], [np.nan, 0.0]]])
    noise = op.noisemap_from_rms(data, rms=[3.0], exptime=[10.0])
    # sqrt((10*3)^2 + |10*4|)/10, sqrt(900 + 10)/10, NaN, sqrt(900)/10
    assert np.allclose(noise[0, 0], [np.sqrt(940.0) / 10, np.sqrt(910.0) / 10])
    assert np.isnan(noise[0, 1, 0]) and np.isclose(noise[0, 1, 1], 3.0)
    d, s, w, cnt = op.prepare(data, rms=[3.0], exptime=[10.0], coefficient=[2.0], bad=[[[False, True], [False, False]]],
                              nan_noise=1e7, noise_boost=1000.0)
    assert d[0, 1, 0] == 0.0 and s[0, 1, 0] == 1e7 and cnt.tolist() == [2]
    assert np.isclose(d[0, 0, 0], 2.0) and np.isclose(s[0, 0, 1], 1000.0 * np.sqrt(910.0) / 20)
    assert w[0, 0, 1] == 0.0 and w[0, 1, 0] == 0.0 and np.isclose(w[0, 1, 1], 1.0 / 1.5 ** 2)
    # whole-epoch boost applies once, however many pixels are flagged (star_photometry.py:316)
    bad = np.zeros((2, 2, 2), bool)
    bad[1, 0, :] = True
    _, s2, _, _ = op.prepare(np.ones((2, 2, 2)), noisemap=np.ones((2, 2, 2)), bad=bad, noise_boost=1000.0,
                             boost_whole_stamp=True)
    assert np.all(s2[0] == 1.0) and np.all(s2[1] == 1000.0)
