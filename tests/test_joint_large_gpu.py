"""GPU parity of the large-grid joint-fit kernels (spectrum scratch in HBM, multi-block starlet / update;
what n = 128 ROIs -- BASELINE.json configs[4] -- run on).

Two legs: (1) the same kernels forced onto n = 16 / 32 stamps by `lc_joint_set_debug_global`, compared with
the oracle exactly like tests/test_joint_gpu.py; (2) the real n = 128, ss = 2 instantiation (N = 256, FFT
length 512) on a two-epoch problem the fp64 oracle still finishes in seconds.
Tolerances as in test_joint_gpu.py (fp32 FFT convolution vs fp64): 3e-5 models / losses, 1e-4 gradients."""
import numpy as np
import pytest

from oracle import model as om, optim as oo
from tests import helpers as H
from tests.test_joint_gpu import _setup

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['1', '3', None], ids=['one-workgroup', 'three-workgroups', 'default-split'])
def global_kernels(ctx, request, monkeypatch):
    """Small stamps through the global-spectrum kernels, in each launch form: one workgroup per epoch (one kernel), or
    the epoch spread over several workgroups with one launch per phase (LCMI_EPOCH_PARTS; unset: as many as leave no
    CU idle, at most four - what these few-epoch problems get by default)."""
    from lightcurver_amd import _lib
    lib = _lib.lib()
    if request.param is None:
        monkeypatch.delenv('LCMI_EPOCH_PARTS', raising=False)
    else:
        monkeypatch.setenv('LCMI_EPOCH_PARTS', request.param)
    lib.lc_joint_set_debug_global(1)
    yield
    lib.lc_joint_set_debug_global(0)


@pytest.mark.parametrize('E,M,n,alpha', [(4, 2, 16, 0.0), (3, 2, 16, 2.0), (2, 3, 32, 0.3)])
def test_forced_global_loss_and_gradients(ctx, global_kernels, E, M, n, alpha):
    ss = 2
    ds, j, po, data, sig2, psf = _setup(ctx, E, M, n, ss, 300 + n + M, alpha_sigma=alpha)
    W = om.propagate_noise_deconv(sig2, psf, ss)
    lam = dict(lam_scales=1.5, lam_hf=0.8, lam_pos=20.0, lam_pos_ps=5.0, lam_fu=0.7)
    prior = [('c_x', po['c_x'] + 0.05, np.full(M, 0.5)), ('c_y', po['c_y'] - 0.02, np.full(M, 0.7))]
    j.set_loss(W=W.numpy(), lam_scales=1.5, lam_hf=0.8, lam_positivity=20.0, lam_positivity_ps=5.0,
               lam_flux_uniformity=0.7,
               prior=dict(c_x_mean=prior[0][1].numpy(), c_x_sigma=prior[0][2], c_y_mean=prior[1][1].numpy(),
                          c_y_sigma=prior[1][2]))
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean']
    j.set_free(free)
    model, chi2_e = j.model()
    mo = om.deconv_model(po, psf, ss, n)
    assert H.rel_err(model, mo.numpy()) < 3e-5
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, W=W, prior=prior, **lam)
    L, g = oo.value_and_grad(fn, po, free)
    loss, grads = j.loss_grad(free)
    assert abs(loss - L) / abs(L) < 3e-5
    for k in free:
        assert H.rel_err(grads[k], g[k].numpy()) < 1e-4, k
    # without a weight cube the penalty uses the per-scale norms
    j.set_loss(lam_scales=2.0, lam_hf=1.0)
    L2, g2 = oo.value_and_grad(lambda q: om.deconv_loss(q, data, sig2, psf, ss, lam_scales=2.0, lam_hf=1.0), po, ['h'])
    loss2, grads2 = j.loss_grad(['h'])
    assert abs(loss2 - L2) / abs(L2) < 3e-5
    assert H.rel_err(grads2['h'], g2['h'].numpy()) < 1e-4


@pytest.mark.parametrize('n', [16, 32])
def test_forced_global_point_source_starlet_term(ctx, global_kernels, n):
    """regularization_strength_pts_source through the multi-block kernels (what 128 x 128 ROIs use; the reference's
    default ROI fit has the term on, roi_modelling.py:311): loss and gradients, then a trajectory, against the oracle."""
    E, M, ss, T = 3, 2, 2, 15
    ds, j, po, data, sig2, psf = _setup(ctx, E, M, n, ss, 600 + n)
    W = om.propagate_noise_deconv(sig2, psf, ss)
    kw = dict(lam_scales=1.0, lam_hf=1.0, lam_pos=10.0, lam_pts=0.3, lam_fu=0.2)
    j.set_loss(W=W.numpy(), lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0, lam_pts_source=0.3, lam_flux_uniformity=0.2)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean']
    j.set_free(free)
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, W=W, **kw)
    L, g = oo.value_and_grad(fn, po, free)
    loss, grads = j.loss_grad(free)
    assert abs(loss - float(L)) / abs(float(L)) < 3e-5
    for k in free:
        assert H.rel_err(grads[k], g[k].numpy()) < 1e-4, k
    j.run_adabelief(T, init_learning_rate=1e-3, schedule_learning_rate=True)
    hist = j.loss_history()
    pf, lh, l0 = oo.adabelief(fn, po, free, 1e-3, T, schedule=True)
    ref = np.array([l0] + lh)
    assert np.abs(hist - ref).max() / np.abs(ref).max() < 2e-4
    got = j.get_params()
    assert H.rel_err(got['a'], pf['a'].numpy()) < 2e-4
    assert np.abs(got['c_x'] - pf['c_x'].numpy()).max() < 5e-4


@pytest.mark.parametrize('n', [16, 32])
def test_forced_global_adabelief_trajectory(ctx, global_kernels, n):
    E, M, T, ss = 4, 2, 20, 2
    ds, j, po, data, sig2, psf = _setup(ctx, E, M, n, ss, 40 + n)
    W = om.propagate_noise_deconv(sig2, psf, ss)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h']
    j.set_loss(W=W.numpy(), lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0)
    j.set_free(free)
    j.run_adabelief(T, init_learning_rate=1e-3, schedule_learning_rate=True)
    hist = j.loss_history()
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, W=W, lam_scales=1.0, lam_hf=1.0, lam_pos=10.0)
    pf, lh, l0 = oo.adabelief(fn, po, free, 1e-3, T, schedule=True)
    ref = np.array([l0] + lh)
    assert np.abs(hist - ref).max() / np.abs(ref).max() < 2e-4
    got = j.get_params()
    assert H.rel_err(got['a'], pf['a'].numpy()) < 2e-4
    assert np.abs(got['c_x'] - pf['c_x'].numpy()).max() < 5e-4
    dh = np.abs(got['h'] - pf['h'].numpy())
    assert dh.max() < 0.05 * T * 1e-3 and np.median(dh) < 1e-5


def test_n128_loss_gradients_and_steps(ctx):
    """configs[4] stamp size (128 x 128 ROI, ss = 2) with two epochs: loss, every gradient block, and a few
    AdaBelief iterations against the oracle."""
    E, M, n, ss, T = 2, 2, 128, 2, 3
    ds, j, po, data, sig2, psf = _setup(ctx, E, M, n, ss, 777, alpha_sigma=0.2)
    lam = dict(lam_scales=1.0, lam_hf=1.0, lam_pos=10.0, lam_pos_ps=5.0, lam_pts=0.05)
    j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0, lam_positivity_ps=5.0, lam_pts_source=0.05)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean']
    j.set_free(free)
    model, chi2_e = j.model()
    mo = om.deconv_model(po, psf, ss, n)
    assert H.rel_err(model, mo.numpy()) < 3e-5
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, **lam)
    L, g = oo.value_and_grad(fn, po, free)
    loss, grads = j.loss_grad(free)
    assert abs(loss - L) / abs(L) < 3e-5
    for k in free:
        # dx / dy: fp32 sums of 65536 (scene gradient x background slope) products, hence the wider bound
        assert H.rel_err(grads[k], g[k].numpy()) < (3e-4 if k in ('dx', 'dy') else 1e-4), k
    j.run_adabelief(T, init_learning_rate=1e-3, schedule_learning_rate=True)
    hist = j.loss_history()
    pf, lh, l0 = oo.adabelief(fn, po, free, 1e-3, T, schedule=True)
    ref = np.array([l0] + lh)
    assert np.abs(hist - ref).max() / np.abs(ref).max() < 2e-4
    assert H.rel_err(j.get_params()['a'], pf['a'].numpy()) < 2e-4


def test_n128_noise_propagation_device_vs_host(ctx):
    """Noise propagation at the configs[4] stamp size: the device path (FFT pipeline of the large-grid epoch kernel)
    against the library's double-precision host implementation."""
    import os
    ds, j, po, data, sig2, psf = _setup(ctx, 2, 2, 128, 2, 778)
    W = j.propagate_noise()
    os.environ['LCMI_NOISE_HOST'] = '1'
    try:
        Wh = j.propagate_noise()
    finally:
        os.environ.pop('LCMI_NOISE_HOST', None)
    assert W.shape == Wh.shape == (9, 256, 256)
    for s in range(W.shape[0]):
        assert H.rel_err(W[s], Wh[s]) < 2e-4, s


def test_forced_global_noise_propagation(ctx, global_kernels):
    ds, j, po, data, sig2, psf = _setup(ctx, 3, 2, 32, 2, 31)
    W = j.propagate_noise()
    Wo = om.propagate_noise_deconv(sig2, psf, 2).numpy()
    assert H.rel_err(W, Wo) < 1e-4
