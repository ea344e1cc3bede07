"""apply_distortion (kernel K13, csrc/distort.hip) through the STARRED-shaped facade against the float64 oracle
(oracle/model.py apply_distortion; scipy.ndimage.map_coordinates pins the oracle's interpolation in
tests/test_oracle_cpu.py), with the call shapes of the reference (star_photometry.py:291-304,
roi_file_preparation.py:169-180: kwargs read back key by key from the regions file, one rescaled position).
Tolerance: fp32 bilinear resampling vs float64, 2e-6 of the peak."""
import numpy as np
import pytest

from oracle import model as om
from lightcurver_amd.synthetic import make_narrow_psf

pytestmark = pytest.mark.gpu


def _psf(N, ss, seed):
    rng = np.random.default_rng(seed)
    return make_narrow_psf(rng, N, ss)[0]


@pytest.mark.parametrize('N,ss', [(32, 2), (48, 2), (64, 2), (33, 1)])
def test_apply_distortion_matches_oracle(ctx, N, ss):
    from lightcurver_amd.starred.psf.psf import apply_distortion, distortion_coefficients
    psf = _psf(N, ss, 5 + N)
    kw = {'dilation_x': np.array([0.01, 0.04, -0.02]), 'dilation_y': np.array([-0.015, 0.01, 0.03]),
          'shear': np.array([0.004, -0.02, 0.01])}
    coef = distortion_coefficients(kw)
    for xy in ([0.0, 0.0], [0.31, -0.42], [-0.5, 0.5]):
        got = apply_distortion(psf, kw, np.array([xy]), ctx=ctx)       # (1, 2) position as the reference passes it
        ref = om.apply_distortion(psf, coef, xy[0], xy[1]).numpy()
        assert got.shape == (N, N)
        assert np.abs(got - ref).max() < 2e-6 * ref.max()
        assert abs(got.sum() - 1.0) < 1e-5
    many = apply_distortion(psf, kw, np.array([[0.1, 0.2], [-0.3, 0.4], [0.5, -0.5]]), ctx=ctx)
    assert many.shape == (3, N, N)
    assert np.abs(many[1] - om.apply_distortion(psf, coef, -0.3, 0.4).numpy()).max() < 2e-6 * many[1].max()


def test_identity_and_errors(ctx):
    from lightcurver_amd import _lib
    from lightcurver_amd.starred.psf.psf import apply_distortion
    psf = _psf(32, 2, 1)
    same = apply_distortion(psf, {}, np.array([0.2, -0.1]), ctx=ctx)    # empty kwargs: no distortion
    assert np.abs(same - psf / psf.sum()).max() < 1e-7
    zero = apply_distortion(psf, {k: np.zeros(3) for k in ('dilation_x', 'dilation_y', 'shear')}, [0.2, -0.1], ctx=ctx)
    assert np.abs(zero - same).max() == 0.0
    with pytest.raises(KeyError):
        apply_distortion(psf, {'twist': np.zeros(3)}, [0.0, 0.0], ctx=ctx)
    with pytest.raises(_lib.LcError):   # inverting matrix
        apply_distortion(psf, {'dilation_x': np.array([-2.0, 0, 0])}, [0.0, 0.0], ctx=ctx)


# ---- build_psf(field_distortion=True): psf_modelling.py:164-171 with field_distortion / stamp_coordinates ------------
def _distorted_problem(F, S, n, ss, seed):
    from lightcurver_amd.synthetic import make_psf_dataset
    from tests import helpers as H
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=seed)
    rng = np.random.default_rng(seed + 1)
    plist = [H.psf_initial_params(ds, f, ss, rng, 0.2) for f in range(F)]
    xy = rng.uniform(-0.5, 0.5, (F, S, 2))
    dist = rng.uniform(-0.04, 0.04, (F, 9))
    for f in range(F):
        plist[f]['dist'] = om.T(dist[f])
        plist[f]['B'] = om.T(2e-4 * rng.standard_normal(plist[f]['B'].shape))
    return ds, plist, xy, dist


def test_distorted_psf_model_loss_and_gradients_match_oracle(ctx):
    """The two-batch scheme of the distortion fit (include/lcmi.h): resampled grid per star, quadratic-form Moffat per
    star, adjoint resampling.  Loss, model, d/d(a, x0, y0), d chi2/dB and - through the chain rule of the facade - d/d(Moffat
    parameters, 9 distortion coefficients) against torch autograd of oracle/model.py psf_loss_distorted."""
    from oracle import optim as oo
    from lightcurver_amd.psf_batch import PsfBatch
    from lightcurver_amd.starred.procedures.psf_routines import quadratic_forms
    from tests import helpers as H
    F, S, n, ss = 2, 4, 16, 2
    N = n * ss
    ds, plist, xy, dist = _distorted_problem(F, S, n, ss, 31)
    w = H.weights_from(ds)
    star_b = PsfBatch(ds['data'].reshape(F * S, 1, n, n), w.reshape(F * S, 1, n, n), ss, ctx)
    frame_b = PsfBatch(np.zeros((F, 1, n, n)), np.zeros((F, 1, n, n)), ss, ctx)
    theta = np.concatenate([H.moffat_array(plist).astype(np.float64), dist], axis=1)
    theta[:, :4] = [[float(p[k]) for k in ('fwhm_x', 'fwhm_y', 'phi', 'beta')] for p in plist]
    q = quadratic_forms(theta, xy, ss)
    star_b.set_moffat_q(q.reshape(F * S, 4))
    star_b.set_stars(H.stars_array(plist).reshape(F * S, 1, 4))
    star_b.set_regularization(None, 0.0, 0.0)
    frame_b.set_grid(np.stack([p['B'].numpy() for p in plist]))
    frame_b.set_distortion(S, dist, xy)
    frame_b.distortion_forward(star_b)
    out = star_b.evaluate(model=True)
    frame_b.distortion_backward(star_b)
    gB = frame_b.get_ext_grad()
    free = ['fwhm_x', 'fwhm_y', 'phi', 'beta', 'dist', 'a', 'x0', 'y0', 'B']
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        fn = lambda p_: om.psf_loss_distorted(p_, xy[f], data, sig2, mask, ss)
        L, g = oo.value_and_grad(fn, plist[f], free)
        loss = out['loss'].reshape(F, S)[f].sum()
        assert abs(loss - L) / abs(L) < 2e-5
        mo = om.psf_model_distorted(plist[f], xy[f], ss, n).numpy()
        assert H.rel_err(out['model'].reshape(F, S, n, n)[f], mo) < 2e-5
        gs = out['grad_stars'].reshape(F, S, 4)[f]
        for k, name in enumerate(('a', 'x0', 'y0')):
            assert H.rel_err(gs[:, k], g[name].numpy()) < 5e-5, name
        assert H.rel_err(gB[f], g['B'].numpy().reshape(N, N)) < 5e-5
    # chain rule theta -> q (what the facade's stage A does: B = 0 there), against autograd of the distorted Moffat
    frame_b.set_grid(None)
    frame_b.distortion_forward(star_b)
    out0 = star_b.evaluate()
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        p0 = dict(plist[f], B=om.T(np.zeros(N * N)))
        L, g = oo.value_and_grad(lambda p_: om.psf_loss_distorted(p_, xy[f], data, sig2, mask, ss), p0,
                                 ['fwhm_x', 'fwhm_y', 'phi', 'beta', 'dist'])
        assert abs(out0['loss'].reshape(F, S)[f].sum() - L) / abs(L) < 2e-5
        gq = out0['grad_moffat'].astype(np.float64).reshape(F, S, 4)[f]
        gt = np.zeros(13)
        for k in range(13):
            h = 1e-6
            tp, tm = theta[f:f + 1].copy(), theta[f:f + 1].copy()
            tp[0, k] += h
            tm[0, k] -= h
            dq = (quadratic_forms(tp, xy[f:f + 1], ss) - quadratic_forms(tm, xy[f:f + 1], ss))[0] / (2 * h)
            gt[k] = (gq * dq).sum()
        ref = np.concatenate([[float(g[k]) for k in ('fwhm_x', 'fwhm_y', 'phi', 'beta')], g['dist'].numpy()])
        assert H.rel_err(gt[:4], ref[:4]) < 2e-4
        assert H.rel_err(gt[4:], ref[4:]) < 2e-4
    star_b.close()
    frame_b.close()


def test_build_psf_with_field_distortion(ctx):
    """Stamps drawn from a PSF that is dilated / sheared across the field (oracle forward model + noise): the fit with
    field_distortion=True must describe them better than the fit without, return kwargs_distortion in the layout the
    reference stores (psf_modelling.py:199-201) and that apply_distortion consumes (star_photometry.py:293-304), and
    recover the sign and size of the dominant coefficient."""
    from lightcurver_amd.starred.procedures.psf_routines import build_psf
    from lightcurver_amd.starred.psf.psf import apply_distortion
    S, n, ss = 8, 24, 2
    N = n * ss
    rng = np.random.default_rng(77)
    xy = rng.uniform(-0.5, 0.5, (S, 2))
    true = np.array([0.0, 0.16, 0.0, 0.0, 0.0, -0.12, 0.0, 0.0, 0.0])   # dilation_x grows with x, dilation_y shrinks with y
    p = dict(fwhm_x=om.T(3.2), fwhm_y=om.T(3.0), phi=om.T(0.3), beta=om.T(3.0), B=om.T(np.zeros(N * N)),
             a=om.T(rng.uniform(2e5, 6e5, S)), x0=om.T(rng.uniform(-0.4, 0.4, S)), y0=om.T(rng.uniform(-0.4, 0.4, S)),
             sky=om.T(np.zeros(S)), dist=om.T(true))
    clean = om.psf_model_distorted(p, xy, ss, n).numpy()
    noise = np.sqrt(5.0 ** 2 + np.abs(clean))
    data = clean + noise * rng.standard_normal(clean.shape)
    kw = dict(image=data, noisemap=noise, subsampling_factor=ss, masks=np.ones_like(data), n_iter_analytic=150,
              n_iter_adabelief=300, guess_method_star_position='center', guess_fwhm_pixels=3.5)
    plain = build_psf(field_distortion=False, **kw)
    res = build_psf(field_distortion=True, stamp_coordinates=xy, **kw)
    kd = res['kwargs_psf']['kwargs_distortion']
    assert set(kd) == {'dilation_x', 'dilation_y', 'shear'} and all(np.asarray(v).shape == (3,) for v in kd.values())
    assert len(res['adabelief_extra_fields']['loss_history']) == 300 and res['residuals'].shape == data.shape
    assert res['narrow_psf'].shape == (N, N) and abs(res['narrow_psf'].sum() - 1.0) < 1e-4
    print('chi2 plain', plain['chi2'], 'with distortion', res['chi2'], 'coefficients', kd)
    assert res['chi2'] < 0.8 * plain['chi2'] and res['chi2'] < 1.5
    assert abs(kd['dilation_x'][1] - 0.16) < 0.05 and abs(kd['dilation_y'][2] + 0.12) < 0.05
    star_psf = apply_distortion(res['narrow_psf'], kd, xy[:1], ctx=ctx)   # the star-photometry call shape
    assert star_psf.shape == (N, N) and abs(star_psf.sum() - 1.0) < 1e-5


def test_build_psf_with_field_distortion_at_a_size_without_a_kernel(ctx):
    """stamp_size_stars is a free integer in the reference's configuration (config.yaml:205): 20 x 20 stamps have no kernel of
    their own and are fitted embedded in 24 x 24 frames (zero-weight ring), since round 4 with field_distortion=True too - the
    two batches of the distortion fit run at the fitted size, every output comes back at the caller's."""
    from lightcurver_amd.starred.procedures.psf_routines import build_psf
    S, n, ss = 8, 20, 2
    N = n * ss
    rng = np.random.default_rng(78)
    xy = rng.uniform(-0.5, 0.5, (S, 2))
    true = np.array([0.0, 0.16, 0.0, 0.0, 0.0, -0.12, 0.0, 0.0, 0.0])
    p = dict(fwhm_x=om.T(3.0), fwhm_y=om.T(2.8), phi=om.T(0.3), beta=om.T(3.0), B=om.T(np.zeros(N * N)),
             a=om.T(rng.uniform(2e5, 6e5, S)), x0=om.T(rng.uniform(-0.4, 0.4, S)), y0=om.T(rng.uniform(-0.4, 0.4, S)),
             sky=om.T(np.zeros(S)), dist=om.T(true))
    clean = om.psf_model_distorted(p, xy, ss, n).numpy()
    noise = np.sqrt(5.0 ** 2 + np.abs(clean))
    data = clean + noise * rng.standard_normal(clean.shape)
    kw = dict(image=data, noisemap=noise, subsampling_factor=ss, masks=np.ones_like(data), n_iter_analytic=150,
              n_iter_adabelief=300, guess_method_star_position='center', guess_fwhm_pixels=3.3)
    plain = build_psf(field_distortion=False, **kw)
    res = build_psf(field_distortion=True, stamp_coordinates=xy, **kw)
    kd = res['kwargs_psf']['kwargs_distortion']
    assert res['residuals'].shape == data.shape and res['models'].shape == data.shape
    assert res['narrow_psf'].shape == (N, N) and abs(res['narrow_psf'].sum() - 1.0) < 1e-4
    assert res['full_psf'].shape == (N, N)
    assert np.asarray(res['kwargs_psf']['kwargs_background']['background']).shape == (N * N,)
    print('chi2 plain', plain['chi2'], 'with distortion', res['chi2'], 'coefficients', kd)
    assert res['chi2'] < 0.8 * plain['chi2'] and res['chi2'] < 1.5
    assert abs(kd['dilation_x'][1] - 0.16) < 0.06 and abs(kd['dilation_y'][2] + 0.12) < 0.06
