"""GPU parity of the PSF-fit kernels (through the C ABI) against the float64 oracle.

Tolerances: the HIP path computes in fp32, the oracle in fp64.  Single evaluations (loss, model,
gradients) must agree to 2e-5 relative to the largest element; short AdaBelief trajectories to
1e-4 on the loss history; moved pixels are bounded as explained in the test (sign-like first
steps amplify fp32 rounding of near-zero gradients); converged fits are checked in test_build_psf_gpu.py.
"""
import numpy as np
import pytest
import torch

from oracle import model as om, optim as oo
from lightcurver_amd.synthetic import make_psf_dataset
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _setup(n, ss, F, S, seed, ctx, jitter=0.3):
    from lightcurver_amd.psf_batch import PsfBatch
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=seed)
    rng = np.random.default_rng(seed + 1)
    plist = [H.psf_initial_params(ds, f, ss, rng, jitter) for f in range(F)]
    b = PsfBatch(ds['data'], H.weights_from(ds), ss, ctx)
    b.set_moffat(H.moffat_array(plist))
    b.set_stars(H.stars_array(plist))
    b.set_grid(np.stack([p['B'].numpy() for p in plist]))
    return ds, plist, b


@pytest.mark.parametrize('n,ss,S', [(16, 1, 3), (16, 2, 4), (24, 2, 5), (32, 2, 8), (64, 2, 8)])
def test_eval_matches_oracle(ctx, n, ss, S):
    # (64, 2, 8) is the instantiation BASELINE.json configs[2] (C3) runs: psf_fit_kernel<PsfCfg<128, ...>>
    F = 2
    ds, plist, b = _setup(n, ss, F, S, 11 + n + ss, ctx)
    N = n * ss
    J = om.n_scales(N)
    Ws = []
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        Ws.append(om.propagate_noise_psf(plist[f], sig2, mask, ss))
    b.set_regularization(np.stack([w[:J].numpy() for w in Ws]), lam_scales=1.3, lam_hf=0.7)
    out = b.evaluate(model=True)
    free = ['fwhm_x', 'fwhm_y', 'phi', 'beta', 'a', 'x0', 'y0', 'sky', 'B']
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=Ws[f], lam_scales=1.3, lam_hf=0.7)
        L, g = oo.value_and_grad(fn, plist[f], free)
        model = om.psf_model(plist[f], ss, n).numpy()
        assert H.rel_err(out['model'][f], model) < 2e-5
        assert abs(out['loss'][f] - L) / abs(L) < 2e-5
        gs = np.stack([g['a'].numpy(), g['x0'].numpy(), g['y0'].numpy(), g['sky'].numpy()], axis=-1)
        for q in range(4):
            assert H.rel_err(out['grad_stars'][f][:, q], gs[:, q]) < 5e-5, q
        assert H.rel_err(out['grad_grid'][f], g['B'].numpy().reshape(N, N)) < 5e-5
        gm = np.array([float(g[k]) for k in ['fwhm_x', 'fwhm_y', 'phi', 'beta']])
        assert H.rel_err(out['grad_moffat'][f], gm) < 1e-4


def test_eval_without_weights_uses_scale_norms(ctx):
    n, ss, S, F = 16, 2, 3, 1
    ds, plist, b = _setup(n, ss, F, S, 3, ctx)
    b.set_regularization(None, lam_scales=2.0, lam_hf=0.5)
    out = b.evaluate()
    data, sig2, mask = H.psf_oracle_inputs(ds, 0, ss)
    fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=None, lam_scales=2.0, lam_hf=0.5)
    L, g = oo.value_and_grad(fn, plist[0], ['B'])
    assert abs(out['loss'][0] - L) / abs(L) < 2e-5
    assert H.rel_err(out['grad_grid'][0], g['B'].numpy().reshape(n * ss, n * ss)) < 5e-5


@pytest.mark.parametrize('n,ss,S', [(16, 1, 3), (16, 2, 4), (24, 2, 5), (32, 2, 8), (64, 2, 3)])
def test_noise_propagation_matches_oracle(ctx, n, ss, S):
    """Device noise propagation (csrc/psf_noise.h: separable starlet tables + rank-1 products, fp32) against the
    oracle's direct float64 formula, and against the library's independent host implementation (double-precision
    FFT convolutions, LCMI_NOISE_HOST=1).  A ragged frame (zero-weight padding star) is included."""
    import os
    F = 2
    ds, plist, b = _setup(n, ss, F, S, 21 + n, ctx)
    b.propagate_noise()
    W = b.get_weights()
    J = om.n_scales(n * ss)
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        Wo = om.propagate_noise_psf(plist[f], sig2, mask, ss)[:J].numpy()
        assert H.rel_err(W[f], Wo) < 2e-5
    os.environ['LCMI_NOISE_HOST'] = '1'
    try:
        b.propagate_noise()
    finally:
        os.environ.pop('LCMI_NOISE_HOST', None)
    assert H.rel_err(W, b.get_weights()) < 2e-5
    # padding star: amplitude 0 and weight 0 -> no contribution, same maps as the frame without it
    from lightcurver_amd.psf_batch import PsfBatch
    w = H.weights_from(ds)
    stars = H.stars_array(plist)
    w2 = np.concatenate([w, np.zeros_like(w[:, :1])], axis=1)
    d2 = np.concatenate([ds['data'], np.zeros_like(ds['data'][:, :1])], axis=1)
    st2 = np.concatenate([stars, np.zeros_like(stars[:, :1])], axis=1)
    b2 = PsfBatch(d2, w2, ss, ctx)
    b2.set_moffat(H.moffat_array(plist))
    b2.set_stars(st2)
    b2.propagate_noise()
    assert H.rel_err(b2.get_weights(), W) < 1e-6


@pytest.mark.parametrize('n,ss,S', [(16, 2, 4), (32, 2, 8), (64, 2, 8)])
def test_adabelief_trajectory_matches_oracle(ctx, n, ss, S):
    F, T = 2, 25
    ds, plist, b = _setup(n, ss, F, S, 5 + n, ctx, jitter=0.1)
    N = n * ss
    J = om.n_scales(N)
    Ws = []
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        Ws.append(om.propagate_noise_psf(plist[f], sig2, mask, ss))
    b.set_regularization(np.stack([w[:J].numpy() for w in Ws]), lam_scales=1.0, lam_hf=1.0)
    b.run_adabelief(T, init_learning_rate=1e-4, schedule_learning_rate=True)
    hist = b.loss_history()
    stars = b.get_stars()
    grid = b.get_grid()
    assert hist.shape == (F, T + 1)
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=Ws[f], lam_scales=1.0, lam_hf=1.0)
        pf, lh, l0 = oo.adabelief(fn, plist[f], ['B', 'a', 'x0', 'y0'], 1e-4, T, schedule=True)
        ref = np.array([l0] + lh)
        assert np.abs(hist[f] - ref).max() / np.abs(ref).max() < 1e-4
        # AdaBelief's first steps are sign-like (|update| ~ lr whatever |g|), so a pixel whose tiny
        # gradient rounds differently in fp32 may drift by a fraction of a step: bound the worst pixel
        # by 2 % of the maximum travel T * lr and require the typical pixel to agree to 1e-7.
        dB = np.abs(grid[f].ravel() - pf['B'].numpy())
        assert dB.max() < 0.02 * T * 1e-4
        assert np.median(dB) < 1e-7
        assert (dB > 1e-6).mean() < 0.01
        assert H.rel_err(stars[f][:, 0], pf['a'].numpy()) < 1e-5
        assert np.abs(stars[f][:, 1] - pf['x0'].numpy()).max() < 2e-5


def test_moffat_stage_reaches_oracle_optimum(ctx):
    """Stage A (Moffat + a, x0, y0 by bounded L-BFGS): the batched device-driven L-BFGS and scipy's
    L-BFGS-B on the oracle loss must land on the same optimum (iterates differ by construction)."""
    import math
    n, ss, S, F = 16, 2, 4, 2
    ds, plist, b = _setup(n, ss, F, S, 31, ctx, jitter=0.0)
    b.set_grid(None)
    stars = H.stars_array(plist)
    stars[..., 3] = 0.0
    b.set_stars(stars)
    final = b.fit_moffat(200)
    mof = b.get_moffat()
    st = b.get_stars()
    free = ['fwhm_x', 'fwhm_y', 'phi', 'beta', 'a', 'x0', 'y0']
    bounds = dict(fwhm_x=(0.5 / ss, n / 2), fwhm_y=(0.5 / ss, n / 2), phi=(-math.pi, math.pi), beta=(1.1, 50.),
                  a=(0, np.inf), x0=(-n / 4, n / 4), y0=(-n / 4, n / 4))
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        p0 = dict(plist[f])
        p0['B'] = torch.zeros_like(p0['B'])
        p0['sky'] = torch.zeros_like(p0['sky'])
        fn = lambda q: om.psf_loss(q, data, sig2, mask, ss)
        po, hist, res = oo.lbfgsb(fn, p0, free, 200, bounds)
        assert final[f] <= res.fun * (1 + 2e-4) + 1e-6, (final[f], res.fun)
        assert abs(final[f] - res.fun) / res.fun < 2e-3
        assert H.rel_err(st[f][:, 0], po['a'].numpy()) < 5e-3
        assert np.abs(st[f][:, 1] - po['x0'].numpy()).max() < 5e-3
        # mean FWHM is well constrained; beta / ellipticity less so
        fw_gpu = 0.5 * (mof[f, 0] + mof[f, 1])
        fw_or = 0.5 * (float(po['fwhm_x']) + float(po['fwhm_y']))
        assert abs(fw_gpu - fw_or) / fw_or < 2e-2


def test_moffat_stage_on_the_device_equals_the_host_driven_one(ctx):
    """Stage A has its L-BFGS state machine on the device (csrc/psf_lbfgs.h, default) and on the host (csrc/lbfgs_host.h,
    LCMI_LBFGS_HOST=1): the same algorithm in double precision on the same fp32 loss / gradient evaluations.  The two
    follow the same iterates up to the order of the dot products, so the optima agree far inside the optimum-level
    tolerance that is asked against scipy."""
    import os
    n, ss, S, F = 32, 2, 8, 12
    out = []
    for env in (None, '1'):
        if env:
            os.environ['LCMI_LBFGS_HOST'] = env
        try:
            ds, plist, b = _setup(n, ss, F, S, 37, ctx, jitter=0.0)
            b.set_grid(None)
            stars = H.stars_array(plist)
            stars[..., 3] = 0.0
            stars[..., 0] *= 0.8          # start away from the optimum
            b.set_stars(stars)
            mof0 = b.get_moffat().copy()
            mof0[:, :2] *= 1.3
            b.set_moffat(mof0)
            out.append((np.asarray(b.fit_moffat(100), dtype=np.float64), b.get_moffat().copy(), np.array(b.get_stars())))
        finally:
            os.environ.pop('LCMI_LBFGS_HOST', None)
    (fd, md, sd), (fh, mh, sh) = out
    assert np.all(np.isfinite(fd)) and np.max(np.abs(fd - fh) / fh) < 1e-4
    assert np.max(np.abs(sd[..., 0] - sh[..., 0]) / np.abs(sh[..., 0])) < 2e-3
    assert np.max(np.abs(sd[..., 1:3] - sh[..., 1:3])) < 2e-3
    assert np.max(np.abs(0.5 * (md[:, 0] + md[:, 1]) - 0.5 * (mh[:, 0] + mh[:, 1]))) < 1e-2


@pytest.mark.parametrize('n,S,F', [(16, 4, 5), (24, 5, 3), (32, 8, 100), (64, 8, 63)])
def test_two_workgroup_form_is_bit_identical(ctx, n, S, F):
    """Small batches run the optimisation loop with two workgroups per frame (chi2 gradient / starlet term,
    swapped through L2 every iteration, psf_kernels.h SPLIT).  Both forms apply the same operations in the
    same order, so loss history, grid and star parameters must agree bit for bit with the one-workgroup form
    (LCMI_PSF_SINGLE_WG=1); F = 100 loads 200 CUs at once, the uneven-load case for the hand-off."""
    import os
    ss, T = 2, (60 if n < 64 else 20)   # n = 64 (one GPU's share of C3): pixel state in HBM, role 1 on its own copy
    out = []
    # the two-workgroup form with its same-XCD hand-off (partners that share an L2: plain stores), with the write-through
    # hand-off everywhere (LCMI_PSF_XCD_FAST=0), and the one-workgroup form
    for env in ({}, {'LCMI_PSF_XCD_FAST': '0'}, {'LCMI_PSF_SINGLE_WG': '1'}):
        os.environ.update(env)
        try:
            ds, plist, b = _setup(n, ss, F, S, 900 + n, ctx, jitter=0.1)
            b.propagate_noise()
            b.set_regularization(None, 1.0, 1.0)
            b.run_adabelief(T, init_learning_rate=1e-4, schedule_learning_rate=True)
            b.run_adabelief(T // 2, init_learning_rate=1e-4, schedule_learning_rate=True)
            out.append((b.loss_history(), b.get_grid(), b.get_stars()))
        finally:
            for k in env:
                os.environ.pop(k, None)
    for other in out[1:]:
        for a, c in zip(out[0], other):
            np.testing.assert_array_equal(a, c)
    assert np.all(np.isfinite(out[0][0]))


def test_two_workgroup_hand_off_under_streaming_load(ctx):
    """The in-launch hand-off of the two-workgroup form must stay exact while other work streams through the
    memory system of every CU: a second context copies 1 GiB buffers in a loop on its own stream (its workgroups
    share the CUs with the partner workgroups, warm their L1/L2 and delay them unevenly) while C2-sized batches are
    fitted; every loss, grid pixel and star parameter must equal the unloaded one-workgroup result bit for bit."""
    import ctypes as C
    import os
    import threading
    from lightcurver_amd import _lib
    n, S, F, ss, T = 32, 8, 100, 2, 150
    os.environ['LCMI_PSF_SINGLE_WG'] = '1'
    try:
        ds, plist, b = _setup(n, ss, F, S, 4321, ctx, jitter=0.1)
        b.propagate_noise()
        b.set_regularization(None, 1.0, 1.0)
        b.run_adabelief(T, init_learning_rate=1e-4)
        ref = (b.loss_history(), b.get_grid(), b.get_stars())
    finally:
        os.environ.pop('LCMI_PSF_SINGLE_WG', None)
    ctx2 = _lib.Context(0)
    stop = threading.Event()
    rates = []

    def stream():
        g = C.c_float()
        while not stop.is_set():
            ctx2.check(_lib.lib().lc_copy_bandwidth(ctx2.h, 1 << 30, 4, C.byref(g)), 'lc_copy_bandwidth')
            rates.append(g.value)

    th = threading.Thread(target=stream)
    th.start()
    try:
        for rep in range(3):
            ds, plist, b = _setup(n, ss, F, S, 4321, ctx, jitter=0.1)
            b.propagate_noise()
            b.set_regularization(None, 1.0, 1.0)
            b.run_adabelief(T, init_learning_rate=1e-4)
            got = (b.loss_history(), b.get_grid(), b.get_stars())
            for a, c in zip(got, ref):
                np.testing.assert_array_equal(a, c)
    finally:
        stop.set()
        th.join()
        ctx2.close()
    assert len(rates) >= 1


@pytest.mark.parametrize('n,S,F', [(32, 8, 12), (64, 4, 5)])
def test_two_workgroup_launch_that_gives_up_is_redone_by_the_library(ctx, n, S, F):
    """A partner workgroup that stops showing up (forced here in role 1 of frame 0 at iteration 7 of the second launch:
    LCMI_PSF_FORCE_ABORT, a test hook) makes the launch give up; the fall-back launch the library enqueues behind every
    two-workgroup launch restores the pre-launch state and redoes it in the one-workgroup form.  Loss history, grid and
    star parameters must equal the one-workgroup result bit for bit, no error is raised, later launches keep working,
    and the batch reports how many launches were redone."""
    import os
    ss, T = 2, 20
    os.environ['LCMI_PSF_SINGLE_WG'] = '1'
    try:
        ds, plist, b = _setup(n, ss, F, S, 700 + n, ctx, jitter=0.1)
        b.propagate_noise()
        b.set_regularization(None, 1.0, 1.0)
        for _ in range(3):
            b.run_adabelief(T, init_learning_rate=1e-4, schedule_learning_rate=True)
        ref = (b.loss_history(), b.get_grid(), b.get_stars())
        assert b.split_fallbacks == 0
    finally:
        os.environ.pop('LCMI_PSF_SINGLE_WG', None)
    ds, plist, b = _setup(n, ss, F, S, 700 + n, ctx, jitter=0.1)
    b.propagate_noise()
    b.set_regularization(None, 1.0, 1.0)
    b.run_adabelief(T, init_learning_rate=1e-4, schedule_learning_rate=True)
    os.environ['LCMI_PSF_FORCE_ABORT'] = '7'
    try:
        b.run_adabelief(T, init_learning_rate=1e-4, schedule_learning_rate=True)
    finally:
        os.environ.pop('LCMI_PSF_FORCE_ABORT', None)
    b.run_adabelief(T, init_learning_rate=1e-4, schedule_learning_rate=True)   # a later launch is not affected
    got = (b.loss_history(), b.get_grid(), b.get_stars())
    assert b.split_fallbacks == 1
    for a, c in zip(got, ref):
        np.testing.assert_array_equal(a, c)
