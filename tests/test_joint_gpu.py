"""GPU parity of the joint (Deconv) forward model, loss, gradients, noise propagation, Fisher
diagonal and AdaBelief trajectories against the float64 oracle, through the C ABI.

Tolerances: fp32 device (FFT convolution) vs fp64 oracle: 3e-5 on models/losses, 1e-4 on
gradients relative to their largest element."""
import numpy as np
import pytest
import torch

from oracle import model as om, optim as oo
from lightcurver_amd.synthetic import make_roi_dataset
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _params(ds, rng, jitter=True, with_h=True):
    t = ds['truth']
    E = ds['data'].shape[0]
    p = {k: np.array(v, dtype=np.float64) for k, v in t.items()}
    if jitter:
        p['a'] = p['a'] * rng.uniform(0.9, 1.1, p['a'].shape)
        p['c_x'] = p['c_x'] + rng.normal(0, 0.1, p['c_x'].shape)
        p['c_y'] = p['c_y'] + rng.normal(0, 0.1, p['c_y'].shape)
        p['dx'] = p['dx'] + rng.normal(0, 0.05, E)
        p['dy'] = p['dy'] + rng.normal(0, 0.05, E)
        p['mean'] = rng.normal(0, 1e-3, E)
        # additive noise: an exactly flat region has starlet coefficients at +-0 whose l1 sub-gradient
        # sign is rounding noise in both the fp32 and the fp64 implementation
        p['h'] = p['h'] * rng.uniform(0.8, 1.2, p['h'].shape) + 2e-3 * rng.standard_normal(p['h'].shape)
    if not with_h:
        p['h'] = np.zeros_like(p['h'])
    return p


def _setup(ctx, E, M, n, ss, seed, alpha_sigma=0.0, with_h=True):
    from lightcurver_amd.joint import JointFit
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=seed, alpha_sigma=alpha_sigma)
    rng = np.random.default_rng(seed + 7)
    p = _params(ds, rng, with_h=with_h)
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, ctx)
    j.set_params(**p)
    po = {k: om.T(v) for k, v in p.items()}
    data, sig2, psf = om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])
    return ds, j, po, data, sig2, psf


@pytest.mark.parametrize('E,M,n,ss,alpha', [(3, 1, 16, 1, 0.0), (4, 2, 16, 2, 0.0), (3, 2, 16, 2, 2.0),
                                            (3, 3, 24, 2, 0.5), (2, 2, 32, 2, 0.3), (2, 2, 64, 2, 0.3),
                                            (3, 2, 64, 2, 0.0),
                                            # stamp sizes beside the tuned ones (config.yaml:205-206 leaves them free):
                                            # the same kernels at that N, multi-block regulariser / update
                                            (2, 2, 40, 2, 0.3), (2, 1, 48, 2, 0.0), (2, 2, 56, 2, 0.0)])
def test_model_loss_and_gradients(ctx, E, M, n, ss, alpha, monkeypatch):
    # n = 64 is the instantiation BASELINE.json configs[3] (C4) runs: joint_epoch_kernel<JointCfg<128,2,192,...>> +
    # joint_update_kernel<128,16>; alpha = 0.3 exercises the ordered-gather T^T, alpha = 0 the 4-tap translation path
    monkeypatch.delenv('LCMI_N128_SPLIT', raising=False)
    _check_model_loss_and_gradients(ctx, E, M, n, ss, alpha)


@pytest.mark.parametrize('parts', ['1', '4', '8', None])
@pytest.mark.parametrize('E,M,alpha', [(3, 2, 0.0), (2, 2, 0.3)])
def test_n64_epoch_spread_over_workgroups(ctx, monkeypatch, parts, E, M, alpha):
    """BASELINE.json configs[3] sharded over GPUs leaves 25 - 100 epochs per GPU.  The 64 x 64 fit can spread an epoch
    over several 4-wave workgroups, one launch per phase, spectrum in global memory (LCMI_N128_SPLIT=1; csrc/joint_fit.hip
    find_jv - measured slower than one workgroup per epoch at every epoch count, so it is not the default;
    LCMI_EPOCH_PARTS forces the count, 1 = one kernel, unset = what the device gets).  Same oracle, same tolerances as the
    one-workgroup kernel."""
    monkeypatch.setenv('LCMI_N128_SPLIT', '1')
    if parts is None:
        monkeypatch.delenv('LCMI_EPOCH_PARTS', raising=False)
    else:
        monkeypatch.setenv('LCMI_EPOCH_PARTS', parts)
    _check_model_loss_and_gradients(ctx, E, M, 64, 2, alpha)


def _check_model_loss_and_gradients(ctx, E, M, n, ss, alpha):
    ds, j, po, data, sig2, psf = _setup(ctx, E, M, n, ss, 100 + n + M, alpha_sigma=alpha)
    N = n * ss
    W = om.propagate_noise_deconv(sig2, psf, ss)
    lam = dict(lam_scales=1.5, lam_hf=0.8, lam_pos=20.0, lam_pos_ps=5.0, lam_fu=0.7, lam_pts=0.3)
    prior = [('c_x', po['c_x'] + 0.05, np.full(M, 0.5)), ('c_y', po['c_y'] - 0.02, np.full(M, 0.7))]
    j.set_loss(W=W.numpy(), lam_scales=1.5, lam_hf=0.8, lam_positivity=20.0, lam_positivity_ps=5.0,
               lam_flux_uniformity=0.7, lam_pts_source=0.3,
               prior=dict(c_x_mean=prior[0][1].numpy(), c_x_sigma=prior[0][2], c_y_mean=prior[1][1].numpy(),
                          c_y_sigma=prior[1][2]))
    j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean'])
    model, chi2_e = j.model()
    mo = om.deconv_model(po, psf, ss, n)
    assert H.rel_err(model, mo.numpy()) < 3e-5
    assert H.rel_err(chi2_e, (((data - mo) ** 2) / sig2).sum((-1, -2)).numpy()) < 3e-5
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, W=W, prior=prior, **lam)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean']
    L, g = oo.value_and_grad(fn, po, free)
    loss, grads = j.loss_grad(free)
    assert abs(loss - L) / abs(L) < 3e-5
    for k in free:
        assert H.rel_err(grads[k], g[k].numpy()) < 1e-4, k
    sc, bg = j.deconvolved(1)
    so, bo = om.deconv_deconvolved(po, 1, N, ss)
    assert H.rel_err(sc, so.numpy()) < 1e-5 and H.rel_err(bg, bo.numpy()) < 1e-5


def test_no_background_path(ctx):
    """h == 0 and fixed (star photometry default, star_photometry.py:74-87): scene has point sources only."""
    ds, j, po, data, sig2, psf = _setup(ctx, 5, 1, 16, 2, 5, with_h=False)
    j.set_loss(lam_scales=3.0, lam_hf=3.0)
    j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy'])
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, 2, lam_scales=3.0, lam_hf=3.0)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy']
    L, g = oo.value_and_grad(fn, po, free)
    loss, grads = j.loss_grad(free)
    assert abs(loss - L) / abs(L) < 3e-5
    for k in free:
        assert H.rel_err(grads[k], g[k].numpy()) < 1e-4, k


@pytest.mark.parametrize('E,n,ss', [(4, 16, 2), (3, 16, 1), (3, 24, 2), (5, 32, 2), (2, 64, 2)])
def test_noise_propagation_and_fisher(ctx, E, n, ss):
    """Device noise propagation (csrc/joint_noise.h, through the FFT pipeline of the epoch kernel, fp32) against the
    oracle's float64 formula and against the library's independent host implementation (LCMI_NOISE_HOST=1).
    Tolerance 1e-4 on W: fp32 FFT convolutions of kappa^2, whose dynamic range over a stamp is ~1e3."""
    import os
    ds, j, po, data, sig2, psf = _setup(ctx, E, 2, n, ss, 11 + n)
    W = j.propagate_noise()
    Wo = om.propagate_noise_deconv(sig2, psf, ss).numpy()
    assert W.shape == Wo.shape
    assert H.rel_err(W, Wo) < 1e-4
    for s in range(W.shape[0]):  # every scale on its own, not only relative to the largest one
        assert H.rel_err(W[s], Wo[s]) < 2e-4, s
    os.environ['LCMI_NOISE_HOST'] = '1'
    try:
        Wh = j.propagate_noise()
    finally:
        os.environ.pop('LCMI_NOISE_HOST', None)
    assert H.rel_err(Wh, Wo) < 3e-5
    s = j.fisher_flux_sigma()
    so = om.fisher_flux_sigma(po, sig2, psf, ss).numpy()
    assert H.rel_err(s, so) < 3e-5


@pytest.mark.parametrize('n,ss,with_h', [(16, 2, True), (32, 2, True), (16, 1, False), (64, 2, True), (40, 2, True), (48, 2, False)])
def test_adabelief_trajectory(ctx, n, ss, with_h):
    E, M, T = 4, 2, 20
    ds, j, po, data, sig2, psf = _setup(ctx, E, M, n, ss, 40 + n, with_h=with_h)
    W = om.propagate_noise_deconv(sig2, psf, ss)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean'] + (['h'] if with_h else [])
    j.set_loss(W=W.numpy(), lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0)
    j.set_free(free)
    j.run_adabelief(T, init_learning_rate=1e-3, schedule_learning_rate=True)
    hist = j.loss_history()
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, W=W, lam_scales=1.0, lam_hf=1.0, lam_pos=10.0)
    pf, lh, l0 = oo.adabelief(fn, po, free, 1e-3, T, schedule=True)
    ref = np.array([l0] + lh)
    assert hist.shape == (T + 1,)
    assert np.abs(hist - ref).max() / np.abs(ref).max() < 2e-4
    got = j.get_params()
    assert H.rel_err(got['a'], pf['a'].numpy()) < 2e-4
    assert np.abs(got['c_x'] - pf['c_x'].numpy()).max() < 5e-4
    assert np.abs(got['dx'] - pf['dx'].numpy()).max() < 5e-4
    if with_h:
        dh = np.abs(got['h'] - pf['h'].numpy())
        assert dh.max() < 0.05 * T * 1e-3 and np.median(dh) < 1e-5


@pytest.mark.parametrize('n,with_h', [(16, True), (32, True), (16, False), (64, True)])
def test_adabelief_trajectory_with_point_source_starlet_term(ctx, n, with_h):
    """regularization_strength_pts_source is on by default in the reference's ROI fit (roi_modelling.py:311).  Inside
    lc_joint_run_adabelief the term is evaluated ahead of the update, on the second stream with the background
    regulariser; the trajectory must still follow the oracle, and a step-by-step drive of the same object (the
    sharded-fit entry points, where the term is evaluated inside the update) must give the same parameters."""
    E, M, T, ss = 4, 2, 20, 2
    ds, j, po, data, sig2, psf = _setup(ctx, E, M, n, ss, 140 + n, with_h=with_h)
    W = om.propagate_noise_deconv(sig2, psf, ss)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean'] + (['h'] if with_h else [])
    kw = dict(W=W.numpy(), lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0, lam_pts_source=0.3, lam_flux_uniformity=0.2)
    j.set_loss(**kw)
    j.set_free(free)
    p0 = j.get_params()
    j.run_adabelief(T, init_learning_rate=1e-3, schedule_learning_rate=True)
    hist = j.loss_history()
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, W=W, lam_scales=1.0, lam_hf=1.0, lam_pos=10.0, lam_pts=0.3,
                                  lam_fu=0.2)
    pf, lh, l0 = oo.adabelief(fn, po, free, 1e-3, T, schedule=True)
    ref = np.array([l0] + lh)
    assert np.abs(hist - ref).max() / np.abs(ref).max() < 2e-4
    got = j.get_params()
    assert H.rel_err(got['a'], pf['a'].numpy()) < 2e-4
    assert np.abs(got['c_x'] - pf['c_x'].numpy()).max() < 5e-4
    # same object, driven step by step
    j.set_params(**p0)
    j.set_free(free)
    for _ in range(T):
        j.step_local()
        j.step_update(init_learning_rate=1e-3, schedule_learning_rate=True)
    got2 = j.get_params()
    assert H.rel_err(got2['a'], got['a']) < 1e-5
    assert np.abs(got2['c_x'] - got['c_x']).max() < 1e-5 and np.abs(got2['c_y'] - got['c_y']).max() < 1e-5


@pytest.mark.parametrize('sigma_rel', [1e-4, 1e-2])
def test_flux_uniformity_with_small_scatter(ctx, sigma_rel):
    """regularization_strength_flux_uniformity = lam * sum_i std_e(a_{e,i}) (jnp.std is two-pass; the reference's ROI
    fit starts with all fluxes of a source equal, roi_modelling.py:236-239,273-276).  With a relative scatter of 1e-4
    the variance <a^2> - <a>^2 cancels to nothing in fp32; the device sums the moments centred on a reference flux
    (csrc/joint_kernels.h, kernel 2), so value and gradient must still follow the oracle."""
    from lightcurver_amd.joint import JointFit
    E, M, n, ss = 24, 2, 16, 2
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=77)
    rng = np.random.default_rng(78)
    p = {k: np.array(v, dtype=np.float64) for k, v in ds['truth'].items()}
    base = p['a'].reshape(E, M).mean(axis=0)
    p['a'] = (base[None, :] * (1.0 + sigma_rel * rng.standard_normal((E, M)))).reshape(-1)
    p['a'] = p['a'].astype(np.float32).astype(np.float64)  # the values the device actually holds
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, ctx)
    j.set_params(**p)
    free = ['a']
    j.set_free(free)
    po = {k: om.T(v) for k, v in p.items()}
    data, sig2, psf = om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])
    out = {}
    for lam in (0.0, 10.0):
        j.set_loss(lam_flux_uniformity=lam)
        L, g = oo.value_and_grad(lambda q: om.deconv_loss(q, data, sig2, psf, ss, lam_fu=lam), po, free)
        loss, grads = j.loss_grad(free)
        out[lam] = (float(L), g['a'].numpy(), loss, grads['a'])
    # the term alone (difference of the two evaluations): value and gradient
    term_o, term_d = out[10.0][0] - out[0.0][0], out[10.0][2] - out[0.0][2]
    g_o, g_d = out[10.0][1] - out[0.0][1], out[10.0][3].astype(np.float64) - out[0.0][3]
    assert term_o > 0
    assert abs(out[10.0][2] - out[10.0][0]) / abs(out[10.0][0]) < 3e-5
    assert H.rel_err(out[10.0][3], out[10.0][1]) < 1e-4
    # the chi2 gradient (~1e2..1e4) dwarfs the term's (lam / (E std) * (a - mean) ~ lam / E): isolate it with its own bound
    assert np.abs(g_d - g_o).max() < 2e-3 * np.abs(g_o).max() + 2e-6 * np.abs(out[0.0][1]).max()


@pytest.mark.parametrize('E,M,n,alpha', [(6, 2, 32, 0.0), (4, 2, 64, 0.0), (3, 3, 32, 2.0)])
def test_loss_and_gradients_against_the_c_port(ctx, E, M, n, alpha):
    """A third implementation beside the HIP path and the torch oracle: oracle/joint_cpu.c in float64 (its own radix-2 FFTs,
    hand-derived adjoints; pinned to oracle/model.py at 1e-9 in tests/test_joint_cpu_port_cpu.py) at sizes the torch
    autograd oracle is slow at - every loss term on, 64 x 64 stamps (the kernel C4 runs) included."""
    from oracle.joint_cpu import JointCpu
    ss = 2
    ds, j, po, data, sig2, psf = _setup(ctx, E, M, n, ss, 300 + n + E, alpha_sigma=alpha)
    lam = dict(lam_scales=1.0, lam_hf=1.0, lam_positivity=50.0, lam_positivity_ps=5.0, lam_pts_source=0.05, lam_flux_uniformity=2.0)
    W = j.propagate_noise()
    j.set_loss(W=W, **lam)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean']
    j.set_free(free)
    loss, g = j.loss_grad(free)
    c = JointCpu(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, double=True)
    c.set_params(**{k: v.numpy() for k, v in po.items()})
    c.set_loss(W=W, **lam)
    Lc, gc = c.eval()
    c.close()
    assert abs(loss - Lc) / abs(Lc) < 3e-5, (loss, Lc)
    for k in free:
        assert H.rel_err(g[k], gc[k]) < (3e-4 if k in ('c_x', 'c_y') else 1e-4), k
    j.close()
