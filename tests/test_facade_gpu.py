"""The drop-in boundary: the reference's own structural test of its STARRED calls
(tests/test_starred_calls/test_starred_calls.py:20-64) restated on the same (seeded) fixture, plus the
ParametersDeconv / Loss / Optimizer behaviours the pipeline relies on."""
from copy import deepcopy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fixture(seed=0):
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(-8, 8), np.arange(-8, 8))
    gauss = np.exp(-0.1 * (x ** 2 + y ** 2))
    data = 0.1 * rng.random((5, 16, 16)) + np.repeat(gauss[None, :, :], repeats=5, axis=0)
    noisemap = 0.1 * np.ones((5, 16, 16))
    psf = np.repeat(gauss[None, :, :], repeats=5, axis=0)
    return data, noisemap, psf


@pytest.mark.parametrize('starlet_bg', [True, False])
def test_do_one_star_forward_modelling_contract(starlet_bg):
    from lightcurver_amd.processes.star_photometry import do_one_star_forward_modelling
    data, noisemap, psf = _fixture()
    d0 = data.copy()
    n_iter = 50
    result = do_one_star_forward_modelling(data, noisemap, psf, 1, n_iter, starlet_global_background=starlet_bg)
    assert isinstance(result, dict)
    for key in ('scale', 'kwargs_final', 'fluxes', 'fluxes_uncertainties', 'chi2', 'chi2_per_frame', 'loss_curve',
                'residuals'):
        assert key in result
    assert isinstance(result['scale'], float) and result['scale'] > 0
    assert isinstance(result['kwargs_final'], dict)
    assert isinstance(result['fluxes'], np.ndarray) and isinstance(result['fluxes_uncertainties'], np.ndarray)
    assert result['fluxes'].ndim == 1 and result['fluxes_uncertainties'].ndim == 1
    assert result['fluxes'].size == result['fluxes_uncertainties'].size == data.shape[0]
    assert isinstance(result['chi2'], float) and result['chi2'] >= 0
    assert isinstance(result['chi2_per_frame'], np.ndarray) and result['chi2_per_frame'].ndim == 1
    assert len(result['chi2_per_frame']) == data.shape[0]
    assert len(result['loss_curve']) == n_iter  # no early stop
    assert result['residuals'].shape == data.shape
    # in-place rescaling by nanmax, as callers of the reference rely on (star_photometry.py:47-49)
    assert np.allclose(data * result['scale'], d0)
    assert np.all(np.isfinite(result['fluxes'])) and np.all(result['fluxes_uncertainties'] > 0)
    assert result['deconvolved_image'].shape == (16, 16) and result['starlet_background'].shape == (16, 16)
    lc = np.array(result['loss_curve'])
    assert np.all(np.isfinite(lc)) and lc[-1] < lc[0]


def test_kwargs_manipulations_and_two_stage_fit():
    """deepcopy / del / in-place *= 0 on kwargs, L-BFGS-B stage then AdaBelief stage (roi_modelling.py:259-334)."""
    from lightcurver_amd.starred.deconvolution.deconvolution import setup_model
    from lightcurver_amd.starred.deconvolution.loss import Loss, Prior
    from lightcurver_amd.starred.deconvolution.parameters import ParametersDeconv
    from lightcurver_amd.starred.optim.optimization import Optimizer
    from lightcurver_amd.starred.utils.noise_utils import propagate_noise
    from lightcurver_amd.synthetic import make_roi_dataset
    ds = make_roi_dataset(E=6, M=2, n=16, ss=2, seed=3)
    data, noise, s = ds['data'].astype(np.float64), ds['noisemap'].astype(np.float64), ds['psf']
    t = ds['truth']
    E, M = 6, 2
    a0 = np.tile(t['a'].reshape(E, M).mean(0) * 0.8, E)
    model, k_init, k_up, k_down, k_fixed = setup_model(data, noise ** 2, s, t['c_x'] + 0.2, t['c_y'] - 0.2, 2, list(a0))
    k_init['kwargs_analytic']['alpha'] = t['alpha']
    k_fixed = deepcopy(k_init)
    del k_fixed['kwargs_analytic']['dx']
    del k_fixed['kwargs_analytic']['dy']
    del k_fixed['kwargs_analytic']['a']
    pars = ParametersDeconv(kwargs_init=k_init, kwargs_fixed=k_fixed, kwargs_up=k_up, kwargs_down=k_down)
    prior = Prior(prior_analytic=[['c_x', t['c_x'] + 0.2, np.array(M * [2.0])], ['c_y', t['c_y'] - 0.2, np.array(M * [2.0])]])
    with pytest.warns(UserWarning, match='lambda is not normalized'):
        loss = Loss(data, model, pars, noise ** 2, prior=prior, regularization_strength_flux_uniformity=1.0)
    optim = Optimizer(loss, pars, method='l-bfgs-b')
    best_fit, logL, extra, runtime = optim.minimize(maxiter=40)
    k1 = deepcopy(pars.best_fit_values(as_kwargs=True))
    assert len(extra['loss_history']) >= 1 and extra['loss_history'][-1] <= extra['loss_history'][0]
    # (with h fixed to zero in this stage the fluxes absorb the background, so no truth comparison here)
    assert np.all(np.isfinite(np.array(k1['kwargs_analytic']['a'])))
    k_fixed = deepcopy(k1)
    for grp, name in (('kwargs_background', 'h'), ('kwargs_background', 'mean'), ('kwargs_analytic', 'a'),
                      ('kwargs_analytic', 'c_x'), ('kwargs_analytic', 'c_y'), ('kwargs_analytic', 'dx'),
                      ('kwargs_analytic', 'dy')):
        del k_fixed[grp][name]
    W = propagate_noise(model, noise, k_init, wavelet_type_list=['starlet'], method='SLIT', num_samples=500, seed=1,
                        likelihood_type='chi2', verbose=False, upsampling_factor=2)[0]
    assert W.shape == (6, 32, 32)
    pars = ParametersDeconv(kwargs_init=k1, kwargs_fixed=k_fixed, kwargs_up=k_up, kwargs_down=k_down)
    loss = Loss(data, model, pars, noise ** 2, regularization_terms='l1_starlet', regularization_strength_scales=1.0,
                regularization_strength_hf=1.0, regularization_strength_positivity=100.0, W=W, prior=prior)
    optim = Optimizer(loss, pars, method='adabelief')
    best_fit, logL, extra, runtime = optim.minimize(max_iterations=120, init_learning_rate=1e-4,
                                                    schedule_learning_rate=False, restart_from_init=False,
                                                    stop_at_loss_increase=False, progress_bar=True,
                                                    return_param_history=True)
    assert len(optim.loss_history) == 120
    k_final = deepcopy(pars.best_fit_values(as_kwargs=True))
    # diagnostics of roi_modelling.py:99-110: zero the background / the fluxes in place and re-model
    only_ps = deepcopy(k_final)
    only_ps['kwargs_background']['h'] *= 0.0
    no_ps = deepcopy(k_final)
    no_ps['kwargs_analytic']['a'] *= 0.0
    m_full, m_ps, m_bg = model.model(k_final), model.model(only_ps), model.model(no_ps)
    assert m_full.shape == data.shape
    assert np.allclose(m_ps + m_bg - np.array(k_final['kwargs_background']['mean'])[:, None, None], m_full, atol=2e-5)
    chi2 = np.nansum((data - m_full) ** 2 / noise ** 2, axis=(1, 2)) / model.image_size ** 2
    assert np.all(chi2 < 3.0)
    hi, bg = model.getDeconvolved(k_final, 0)
    assert hi.shape == (32, 32) and bg.shape == (32, 32)
    from lightcurver_amd.utilities.starred_utilities import get_flux_uncertainties
    sig = get_flux_uncertainties(kwargs=k_final, kwargs_up=k_up, kwargs_down=k_down, data=data, noisemap=noise, model=model)
    assert sig.shape == (E * M,) and np.all(sig > 0)
    # de-interleave per source as roi_modelling.py:462 does
    assert sig[0::M].shape == (E,)


def _small_fit(n_free=('a', 'dx', 'dy')):
    from lightcurver_amd.starred.deconvolution.deconvolution import setup_model
    from lightcurver_amd.starred.deconvolution.loss import Loss
    from lightcurver_amd.starred.deconvolution.parameters import ParametersDeconv
    from lightcurver_amd.synthetic import make_roi_dataset
    ds = make_roi_dataset(E=5, M=1, n=16, ss=2, seed=9, with_background=False)
    data, noise, s = ds['data'].astype(np.float64), ds['noisemap'].astype(np.float64), ds['psf']
    a0 = list(0.8 * np.asarray(ds['truth']['a']))
    model, k_init, k_up, k_down, _ = setup_model(data, noise ** 2, s, np.array([0.]), np.array([0.]), 2, a0)
    fixed = deepcopy(k_init)
    for name in n_free:
        del fixed['kwargs_analytic'][name]
    pars = ParametersDeconv(kwargs_init=k_init, kwargs_fixed=fixed, kwargs_up=k_up, kwargs_down=k_down)
    return model, pars, Loss(data, model, pars, noise ** 2), data, noise, s, k_init


def test_param_history_and_early_stop():
    """return_param_history=True (the reference passes it: star_photometry.py:119, roi_modelling.py:331) returns the
    parameter vector after EVERY update; the chunked drive it needs must not change the trajectory.
    stop_at_loss_increase=True stops at the first increase of the loss after min_iterations."""
    from lightcurver_amd.starred.optim.optimization import Optimizer
    T = 25
    model, pars, loss, *_ = _small_fit()
    opt = Optimizer(loss, pars, method='adabelief')
    best, logL, extra, _ = opt.minimize(max_iterations=T, init_learning_rate=1e-3, schedule_learning_rate=True,
                                        restart_from_init=True, return_param_history=True)
    ph = extra['param_history']
    assert len(ph) == T and all(np.asarray(v).shape == np.asarray(best).shape for v in ph)
    assert np.array_equal(np.asarray(ph[-1]), np.asarray(best))
    assert np.any(np.asarray(ph[0]) != np.asarray(ph[5]))
    hist_chunked = list(opt.loss_history)
    model2, pars2, loss2, *_ = _small_fit()
    opt2 = Optimizer(loss2, pars2, method='adabelief')
    best2, *_ = opt2.minimize(max_iterations=T, init_learning_rate=1e-3, schedule_learning_rate=True, restart_from_init=True)
    assert hist_chunked == list(opt2.loss_history) and np.array_equal(np.asarray(best), np.asarray(best2))  # same bits
    # the history is recorded on the device (one run_adabelief call); it equals the one the host-in-the-loop drive
    # collects with one call and one get_params() per iteration
    model4, pars4, loss4, *_ = _small_fit()
    fit4 = loss4.configure()
    fit4.set_params(**pars4._start)
    fit4.set_free(pars4.free)
    rows = []
    for _ in range(T):
        fit4.run_adabelief(1, init_learning_rate=1e-3, schedule_learning_rate=True)
        rows.append(np.concatenate([fit4.get_params()[k] for k in pars4.free]))
    assert np.array_equal(np.asarray(ph, dtype=np.float32), np.asarray(rows, dtype=np.float32))
    # early stop: a learning rate of 0.3 (fluxes ~10, shifts in pixels) overshoots within a few steps
    model3, pars3, loss3, *_ = _small_fit()
    opt3 = Optimizer(loss3, pars3, method='adabelief')
    opt3.minimize(max_iterations=400, min_iterations=5, init_learning_rate=0.3, schedule_learning_rate=False,
                  restart_from_init=True, stop_at_loss_increase=True)
    lh = np.array(opt3.loss_history)
    # the fit stops AT the first update (after min_iterations) that raised the loss
    assert 5 <= lh.size < 400 and lh[-1] > lh[-2] and np.all(np.diff(lh[4:-1]) <= 0)


def test_param_history_is_copied_only_if_somebody_holds_it():
    """The device-resident history costs a device-to-host copy only when it is read: a history the caller dropped is
    released on the device when the fit next needs the buffer (weak reference in the fit), a history the caller still
    holds comes over before the buffer goes away, and a history that cannot be allocated on the device falls back to the
    host-collected rows instead of failing the fit."""
    from lightcurver_amd.starred.optim.optimization import Optimizer
    from lightcurver_amd import _lib
    T = 8
    kw = dict(max_iterations=T, init_learning_rate=1e-3, schedule_learning_rate=True, restart_from_init=True,
              return_param_history=True)
    model, pars, loss, *_ = _small_fit()
    fit = loss.configure()
    copies = []
    real = fit.param_history
    fit.param_history = lambda *a, **k: (copies.append(1), real(*a, **k))[1]
    opt = Optimizer(loss, pars, method='adabelief')
    best, _, extra, _ = opt.minimize(**kw)
    del extra                                  # nobody holds the history
    fit.set_free(pars.free)                    # the fit needs its buffer again: dropped on the device, no copy
    assert copies == [] and fit._l.lc_joint_param_history_rows(fit.h) == 0
    best, _, extra, _ = opt.minimize(**kw)
    held = extra['param_history']
    fit.set_free(pars.free)                    # held: copied before the buffer goes away
    assert copies == [1] and np.array_equal(np.asarray(held[-1]), np.asarray(best))
    # allocation failure -> host-collected rows, same numbers
    real_begin = fit.param_history_begin
    def failing(capacity):
        raise _lib.LcError('lc_joint_param_history_begin: out of memory (test)')
    fit.param_history_begin = failing
    best2, _, extra2, _ = opt.minimize(**kw)
    fit.param_history_begin = real_begin
    assert np.array_equal(np.asarray(extra2['param_history'], dtype=np.float32), np.asarray(held, dtype=np.float32))
    assert np.array_equal(np.asarray(best2), np.asarray(best))


def test_edited_pixels_are_seen_by_the_device_object():
    """Deconv re-uploads its inputs whenever ANY byte of data / variance changed (full hash), e.g. after masking a
    single pixel between two calls as star_photometry.py:309-316 does."""
    model, pars, loss, data, noise, s, k_init = _small_fit()
    m0 = np.array(model.model(k_init))
    fit0 = model._ensure_fit(data, noise ** 2)
    assert model._ensure_fit(data, noise ** 2) is fit0          # unchanged inputs: same device object
    noise2 = noise.copy()
    noise2[2, 5, 7] *= 1000.0                                    # one pixel, off any sub-sampling grid
    fit1 = model._ensure_fit(data, noise2 ** 2)
    assert fit1 is not fit0
    _, chi2_a = fit1.model()
    fit2 = model._ensure_fit(data, noise ** 2)
    _, chi2_b = fit2.model()
    assert chi2_a[2] != chi2_b[2] and np.array_equal(np.delete(chi2_a, 2), np.delete(chi2_b, 2))
    assert np.array_equal(np.array(model.model(k_init)), m0)


def test_outputs_have_the_shapes_the_plotting_modules_index():
    """lightcurver/plotting/psf_plotting.py:36-105 indexes residuals[i] (2-D per star) and full_psf (2-D) and plots
    loss_curve; joint_modelling_plotting.py:30-95 broadcasts residuals / noisemaps as (E, n, n) and shows a 2-D
    starlet_background and deconvolved_image.  The restated step functions must hand over exactly that."""
    from lightcurver_amd.processes.star_photometry import do_one_star_forward_modelling
    from lightcurver_amd.starred.procedures.psf_routines import build_psf
    from lightcurver_amd.synthetic import make_psf_dataset
    ds = make_psf_dataset(F=1, S=4, n=16, ss=2, seed=12)
    res = build_psf(image=ds['data'][0], noisemap=ds['noisemap'][0], subsampling_factor=2, masks=ds['masks'][0],
                    n_iter_analytic=20, n_iter_adabelief=30, guess_method_star_position='center', guess_fwhm_pixels=3.0)
    assert res['full_psf'].ndim == 2 and res['narrow_psf'].shape == (32, 32)
    assert len(res['residuals']) == 4 and all(np.asarray(r).shape == (16, 16) for r in res['residuals'])
    assert np.asarray(res['adabelief_extra_fields']['loss_history']).shape == (30,)
    assert f"{res['chi2']:.02f}"                                  # psf_modelling.py:224 formats it
    data, noisemap, psf = _fixture()
    out = do_one_star_forward_modelling(data, noisemap, psf, 1, 20)
    assert (out['residuals'] / noisemap).shape == data.shape      # joint_modelling_plotting.py:30-32
    assert out['starlet_background'].ndim == 2 and out['deconvolved_image'].ndim == 2
