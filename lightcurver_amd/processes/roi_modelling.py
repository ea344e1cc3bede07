"""Joint modelling of the ROI cutouts: the arithmetic core of the reference's
lightcurver/processes/roi_modelling.py:198-335 (two-stage fit) and :450-474 (fluxes, uncertainties,
per-frame chi2), restated on the GPU STARRED mirror and taking arrays instead of the HDF5 / WCS /
config plumbing (which stays with the caller: SURVEY.md section 2 rows 6, 16, 17 are out of scope).

Stage 1 (roi_modelling.py:259-282): only the per-epoch translations and fluxes are free, L-BFGS-B,
flux-scatter regularisation.  Stage 2 (:284-335): background (optional), per-epoch constants, fluxes,
astrometry and translations, AdaBelief with the starlet-regularised background.
"""
import warnings
from copy import deepcopy

import numpy as np

from ..starred.deconvolution.deconvolution import setup_model
from ..starred.deconvolution.loss import Loss, Prior
from ..starred.deconvolution.parameters import ParametersDeconv
from ..starred.optim.optimization import Optimizer
from ..starred.utils.noise_utils import propagate_noise
from ..utilities.starred_utilities import get_flux_uncertainties

DEFAULT_REGULARIZATION = {  # code fall-backs of the reference (roi_modelling.py:273,308-312)
    'regularization_scatter_fluxes_pre_optim': 10.0,
    'regularization_strength_scales': 1.0,
    'regularization_strength_hf': 1.0,
    'regularization_strength_positivity': 100.0,
    'regularization_strength_pts_source': 0.01,
    'regularization_scatter_fluxes_main_optim': 10.0,
}


def _median_stack(data):
    """np.nanmedian(data, axis=0); the NaN-free case (the usual one) takes np.median, an order of magnitude faster than
    numpy's median along a strided axis for (epochs, n, n) stacks (a sort of contiguous rows)."""
    data = np.asarray(data)
    if np.isnan(data).any():
        return np.nanmedian(data, axis=0)
    E = data.shape[0]
    s = np.sort(np.ascontiguousarray(data.reshape(E, -1).T), axis=1)     # pixels x epochs, each row contiguous
    return (0.5 * (s[:, (E - 1) // 2] + s[:, E // 2])).reshape(data.shape[1:])


def initial_point_source_fluxes(data, xs, ys, radius):
    """Aperture sums on the median stack (roi_modelling.py:198-204 uses photutils' exact circular
    apertures; this is the pixel-centre version, good enough for a starting point)."""
    stack = _median_stack(data)
    n = stack.shape[0]
    yy, xx = np.mgrid[0:n, 0:n]
    return [float(np.nansum(stack[(xx - x) ** 2 + (yy - y) ** 2 <= radius ** 2])) for x, y in zip(xs, ys)]


def model_roi_cutouts(data, noisemap, psf, subsampling_factor, xs_pixels, ys_pixels, angles_to_north=None,
                      initial_a=None, aperture_radius=3.0, fix_point_source_astrometry=False,
                      starting_background=None, further_optimize_background=True, regularization=None,
                      roi_deconv_translations_iters=300, roi_deconv_all_iters=2000, rescale=True):
    """data, noisemap (E, n, n); psf (E, N, N); xs_pixels, ys_pixels: point-source positions in pixels of
    the first epoch (0-based, as astropy's world_to_pixel returns them).

    Returns a dict: kwargs_final, kwargs_stage1, loss_history (stage 2), loss_history_stage1, scale, model,
    and the pieces needed downstream (model object, kwargs_up / kwargs_down).
    """
    reg = dict(DEFAULT_REGULARIZATION)
    reg.update(regularization or {})
    data = np.array(data, dtype=np.float64)
    noisemap = np.array(noisemap, dtype=np.float64)
    scale = float(np.nanmax(data)) if rescale else 1.0
    data /= scale
    noisemap /= scale
    E, n, _ = data.shape
    xs = np.atleast_1d(np.asarray(xs_pixels, dtype=np.float64))
    ys = np.atleast_1d(np.asarray(ys_pixels, dtype=np.float64))
    M = xs.size
    if angles_to_north is None:
        angles = np.zeros(E)
    else:
        angles = np.asarray(angles_to_north, dtype=np.float64) - float(np.asarray(angles_to_north)[0])
    if initial_a is None:
        initial_a = initial_point_source_fluxes(data, xs, ys, aperture_radius)
    offset = (n - 1) / 2.0  # STARRED's origin is the stamp centre
    c_x0, c_y0 = xs - offset, ys - offset
    model, k_init, k_up, k_down, _ = setup_model(data, noisemap ** 2, psf, c_x0, c_y0, subsampling_factor,
                                                 E * list(initial_a))
    k_init['kwargs_analytic']['alpha'] = angles

    prior = None
    fix_astrometry = fix_point_source_astrometry
    if isinstance(fix_astrometry, float):
        prior = Prior(prior_analytic=[['c_x', c_x0, np.full(M, fix_astrometry)], ['c_y', c_y0, np.full(M, fix_astrometry)]])
    h_fixed_to_start = starting_background is not None
    if h_fixed_to_start:
        k_init['kwargs_background']['h'] = np.asarray(starting_background, dtype=np.float64).ravel() / scale

    # ---- stage 1: translations + fluxes --------------------------------------------------------------
    fixed = deepcopy(k_init)
    for name in ('dx', 'dy', 'a'):
        del fixed['kwargs_analytic'][name]
    pars = ParametersDeconv(kwargs_init=k_init, kwargs_fixed=fixed, kwargs_up=k_up, kwargs_down=k_down)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        loss = Loss(data, model, pars, noisemap ** 2, prior=prior,
                    regularization_strength_flux_uniformity=reg['regularization_scatter_fluxes_pre_optim'])
    optim1 = Optimizer(loss, pars, method='l-bfgs-b')
    optim1.minimize(maxiter=int(roi_deconv_translations_iters))
    k_stage1 = deepcopy(pars.best_fit_values(as_kwargs=True))

    # ---- stage 2: everything (but the rotation) --------------------------------------------------------
    fixed = deepcopy(k_stage1)
    if further_optimize_background:
        del fixed['kwargs_background']['h']
    del fixed['kwargs_background']['mean']
    for name in ('a', 'c_x', 'c_y', 'dx', 'dy'):
        del fixed['kwargs_analytic'][name]
    if isinstance(fix_astrometry, bool) and fix_astrometry:
        fixed['kwargs_analytic']['c_x'] = c_x0
        fixed['kwargs_analytic']['c_y'] = c_y0
    W = propagate_noise(model, noisemap, k_init, wavelet_type_list=['starlet'], method='SLIT', num_samples=500, seed=1,
                        likelihood_type='chi2', verbose=False, upsampling_factor=subsampling_factor)[0]
    pars = ParametersDeconv(kwargs_init=k_stage1, kwargs_fixed=fixed, kwargs_up=k_up, kwargs_down=k_down)
    loss = Loss(data, model, pars, noisemap ** 2, regularization_terms='l1_starlet',
                regularization_strength_scales=reg['regularization_strength_scales'],
                regularization_strength_hf=reg['regularization_strength_hf'],
                regularization_strength_positivity=reg['regularization_strength_positivity'],
                regularization_strength_pts_source=reg['regularization_strength_pts_source'],
                regularization_strength_flux_uniformity=reg['regularization_scatter_fluxes_main_optim'],
                W=W, prior=prior)
    optim2 = Optimizer(loss, pars, method='adabelief')
    optim2.minimize(max_iterations=int(roi_deconv_all_iters), init_learning_rate=1e-4, schedule_learning_rate=False,
                    restart_from_init=False, stop_at_loss_increase=False, progress_bar=True, return_param_history=True)   # as the reference passes it (:331): recorded on the device, copied only if read
    k_final = deepcopy(pars.best_fit_values(as_kwargs=True))
    # position of the sources in the first epoch, back in pixels (roi_modelling.py:339-340)
    x_pix = np.array(k_final['kwargs_analytic']['c_x']) + np.array(k_final['kwargs_analytic']['dx'])[0] + offset
    y_pix = np.array(k_final['kwargs_analytic']['c_y']) + np.array(k_final['kwargs_analytic']['dy'])[0] + offset
    return dict(kwargs_final=k_final, kwargs_stage1=k_stage1, loss_history=optim2.loss_history,
                loss_history_stage1=optim1.loss_history, scale=scale, model=model, kwargs_up=k_up, kwargs_down=k_down,
                data=data, noisemap=noisemap, x_pixels=x_pix, y_pixels=y_pix, W=W)


def model_roi_cutouts_sharded(data, noisemap, psf, subsampling_factor, xs_pixels, ys_pixels, initial_a, scale,
                              group=None, use_peer=False, regularization=None, roi_deconv_translations_iters=300,
                              roi_deconv_all_iters=2000, further_optimize_background=True, ctx=None):
    """The two-stage fit of ``model_roi_cutouts`` with the epochs spread over the ranks of a torch.distributed group (one
    process per GPU): every rank passes ITS epochs - data, noisemap (E_local, n, n), psf (E_local, N, N) - and the same
    ``initial_a`` (M fluxes) and ``scale`` (nanmax of the data of ALL epochs: ``global_scale``).  The reference has no
    counterpart (it keeps all epochs on one device, roi_modelling.py:154-160,213); this is what BASELINE configs[4] (1000
    epochs of 128 x 128 over eight GPUs) runs.

    Stage 1: fluxes and translations by L-BFGS-B (``distributed.sharded_lbfgs``: scipy on every rank, the evaluations sharded);
    noise propagation: every rank its epochs, the levels added in quadrature over the ranks; stage 2: AdaBelief on everything
    but the rotation (``ShardedJointOptimizer.run``: the loop in C++, one all-reduce of the shared block per iteration,
    ``use_peer``: by the one-shot peer-memory kernel).  Returns, on every rank, the parameters of ALL epochs (gathered), the
    loss histories and the Fisher 1-sigma of the fluxes at the final point."""
    import torch
    import torch.distributed as dist
    from ..distributed import PeerGroup, ShardedJointOptimizer, gather_epoch_blocks, host_group
    from ..joint import make_joint_fit
    from .. import _lib
    reg = dict(DEFAULT_REGULARIZATION)
    reg.update(regularization or {})
    data = np.array(data, dtype=np.float64) / scale
    noisemap = np.array(noisemap, dtype=np.float64) / scale
    E, n, _ = data.shape
    xs = np.atleast_1d(np.asarray(xs_pixels, dtype=np.float64))
    ys = np.atleast_1d(np.asarray(ys_pixels, dtype=np.float64))
    M, ss = xs.size, int(subsampling_factor)
    offset = (n - 1) / 2.0
    # (a stamp size without an epoch kernel of its own is fitted embedded in the next one, on every rank alike)
    fit = make_joint_fit(data, noisemap ** 2, psf, ss, M, ctx or _lib.default_context())
    peer = None
    try:
        fit.set_params(a=np.tile(np.asarray(initial_a, np.float64) / scale, E), c_x=xs - offset, c_y=ys - offset, dx=np.zeros(E),
                       dy=np.zeros(E), alpha=np.zeros(E), h=np.zeros((n * ss) ** 2), mean=np.zeros(E))
        peer = PeerGroup(fit, group) if use_peer else None
        opt = ShardedJointOptimizer(fit, group, peer=peer)
        # ---- stage 1: translations + fluxes (roi_modelling.py:259-282) ----------------------------------------------
        fit.set_loss(lam_flux_uniformity=reg['regularization_scatter_fluxes_pre_optim'])
        half = n / 2.0
        hist1, _ = opt.run_lbfgs(['a', 'dx', 'dy'], int(roi_deconv_translations_iters), lower={'a': 0.0, 'dx': -half, 'dy': -half},
                                 upper={'dx': half, 'dy': half})
        # ---- noise levels of the starlet coefficients: the epochs add in quadrature, over the ranks too -----------------
        W2 = torch.from_numpy(np.square(fit.propagate_noise().astype(np.float64)))
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == 'nccl':
                W2 = W2.to(torch.device('cuda', fit.ctx.stream()[1]))
            dist.all_reduce(W2, op=dist.ReduceOp.SUM, group=group)
            W2 = W2.cpu()
        W = np.sqrt(W2.numpy())
        # ---- stage 2: everything but the rotation (roi_modelling.py:284-335) -------------------------------------------
        fit.set_loss(W=W, lam_scales=reg['regularization_strength_scales'], lam_hf=reg['regularization_strength_hf'],
                     lam_positivity=reg['regularization_strength_positivity'],
                     lam_pts_source=reg['regularization_strength_pts_source'],
                     lam_flux_uniformity=reg['regularization_scatter_fluxes_main_optim'])
        fit.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean'] + (['h'] if further_optimize_background else []))
        opt.run(int(roi_deconv_all_iters), init_learning_rate=1e-4, schedule_learning_rate=False)
        fit.ctx.synchronize()
        final = gather_epoch_blocks(fit.get_params(), M, group)
        sigma_loc = np.asarray(fit.fisher_flux_sigma(), np.float64)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            parts = [None] * dist.get_world_size(group)
            dist.all_gather_object(parts, sigma_loc, group=host_group(group))
            sigma = np.concatenate(parts)
        else:
            sigma = sigma_loc
        # (the entry lc_joint_get_loss_history appends for the final parameters is a local evaluation: dropped)
        return dict(flat_final=final, loss_history=np.asarray(fit.loss_history(), np.float64)[:-1], loss_history_stage1=hist1,
                    fluxes_sigma=sigma, scale=scale, W=W)
    finally:
        # (errors of the sharded stages are agreed over the ranks before they are raised - distributed.raise_together - so every
        #  rank arrives here, and the bounded host barrier inside peer.close() finds its partners)
        if peer is not None:
            peer.close()
        fit.close()


def global_scale(data, group=None):
    """nanmax of the data over the epochs of all ranks (what ``model_roi_cutouts`` divides by, roi_modelling.py:176)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(np.nanmax(data))], dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == 'nccl':
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t[0])


def fluxes_from_model(model, kwargs, kwargs_up, kwargs_down, data, noisemap, n_sources, model_scale,
                      normalization_errors):
    """Numeric part of get_fluxes_dataframe_from_model (roi_modelling.py:450-474): per-source light
    curves (a is epoch-major: curve i = a[i::M]), their uncertainties (Fisher 1-sigma compounded with the
    frame normalisation error), residuals and the per-frame reduced chi2."""
    a = np.array(kwargs['kwargs_analytic']['a'])
    sigma_a = np.array(get_flux_uncertainties(kwargs=kwargs, kwargs_up=kwargs_up, kwargs_down=kwargs_down,
                                              data=data, noisemap=noisemap, model=model))
    norm_err = np.asarray(normalization_errors, dtype=np.float64)
    curves, d_curves = [], []
    for i in range(n_sources):
        curve = a[i::n_sources] * model_scale
        photon = sigma_a[i::n_sources] * model_scale
        curves.append(curve)
        d_curves.append(np.sqrt(photon ** 2 + (norm_err * curve) ** 2))
    residuals = data - model.model(kwargs)
    chi2_per_frame = np.nansum(residuals ** 2 / noisemap ** 2, axis=(1, 2)) / model.image_size ** 2
    return dict(fluxes=np.array(curves), d_fluxes=np.array(d_curves), residuals=residuals,
                reduced_chi2=np.array(chi2_per_frame))


# ---- diagnostics of roi_modelling.py:34-125 ---------------------------------------------
def align_data_interpolation(array, starred_kwargs):
    """De-translate and de-rotate every epoch with the fitted dx, dy, alpha (spline interpolation, for
    diagnostics only; reference roi_modelling.py:34-57: shift by (-dy, -dx), then rotate by +alpha degrees)."""
    from scipy.ndimage import rotate, shift
    ka = starred_kwargs['kwargs_analytic']
    return np.array([rotate(shift(img, (-ddy, -ddx)), ang, reshape=False)
                     for img, ddx, ddy, ang in zip(array, ka['dx'], ka['dy'], ka['alpha'])])


def sigma_clipped_weighted_stack(data, noisemap, n_sigma=3.0):
    """Average stack with one pass of rejection around the per-pixel median (threshold n_sigma standard
    deviations) and weights 1 / noise: what the reference obtains from ccdproc's Combiner
    (roi_modelling.py:60-83), written out since ccdproc is not a dependency here."""
    data = np.asarray(data, dtype=np.float64)
    weights = 1.0 / np.asarray(noisemap, dtype=np.float64)
    med = _median_stack(data)
    dev = np.nanstd(data, axis=0)
    keep = np.abs(data - med) <= n_sigma * dev
    keep |= ~np.isfinite(dev)[None]
    w = np.where(keep & np.isfinite(data), weights, 0.0)
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.nansum(w * np.nan_to_num(data), axis=0) / np.sum(w, axis=0)


def stack_data_diagnostic(data, noisemap, starred_kwargs, starred_model):
    """Three aligned stacks: the data, the data minus the point sources, the data minus the background
    (reference roi_modelling.py:86-125).  Two extra forward models on the device, the rest on the host."""
    only_ps = deepcopy(starred_kwargs)
    only_ps['kwargs_background']['h'] = np.array(only_ps['kwargs_background']['h']) * 0.0
    no_ps = deepcopy(starred_kwargs)
    no_ps['kwargs_analytic']['a'] = np.array(no_ps['kwargs_analytic']['a']) * 0.0
    data = np.asarray(data, dtype=np.float64)
    minus_ps = align_data_interpolation(data - starred_model.model(only_ps), only_ps)
    minus_bg = align_data_interpolation(data - starred_model.model(no_ps), no_ps)
    aligned = align_data_interpolation(data, starred_kwargs)
    return {'stack': sigma_clipped_weighted_stack(aligned, noisemap),
            'stack_no_ps': sigma_clipped_weighted_stack(minus_ps, noisemap),
            'stack_no_background': sigma_clipped_weighted_stack(minus_bg, noisemap)}
