"""PSF photometry of one star over all its epochs = one joint forward-model fit, restated from the
reference's lightcurver/processes/star_photometry.py:23-151 on top of the GPU STARRED mirror.  Same
signature, same in-place rescaling of the inputs (callers plot the rescaled arrays, :47-49), same keys
in the returned dictionary (tests/test_starred_calls/test_starred_calls.py:21-64)."""
import numpy as np

from ..starred.deconvolution.deconvolution import setup_model
from ..starred.deconvolution.loss import Loss
from ..starred.deconvolution.parameters import ParametersDeconv
from ..starred.optim.optimization import Optimizer
from ..starred.utils.noise_utils import propagate_noise
from ..utilities.starred_utilities import get_flux_uncertainties


def prepare_star_epochs(data, noisemap, cosmics_mask):
    """Clean-up of one star's epoch stack before the joint fit, as the reference does it at
    star_photometry.py:309-316: pixels where BOTH data and noise are NaN become (0, 1e7); then every EPOCH
    that contains at least one flagged pixel has its whole noise map multiplied by 1000 (the reference
    indexes epochs there, ``noisemap[np.where(~mask)[0]] *= 1000``, once per epoch: SURVEY.md 8(a) row a5).
    cosmics_mask: True where flagged.  Returns (data, noisemap) as new float64 arrays."""
    data = np.array(data, dtype=np.float64)
    noisemap = np.array(noisemap, dtype=np.float64)
    both_nan = np.isnan(data) & np.isnan(noisemap)
    data[both_nan] = 0.0
    noisemap[both_nan] = 1e7
    good = ~np.asarray(cosmics_mask, dtype=bool)
    flagged_epochs = np.unique(np.where(~good)[0])
    noisemap[flagged_epochs] *= 1000.0
    return data, noisemap


def _border_level(stack):
    """Sky estimate: mean over the four borders of the per-epoch median of the outermost row / column."""
    edges = [stack[:, :1, :], stack[:, :, :1], stack[:, -1:, :], stack[:, :, -1:]]
    with np.errstate(all='ignore'):
        level = np.nanmean([np.nanmedian(e, axis=(1, 2)) for e in edges])
    return float(np.nan_to_num(level, nan=0.0))


def do_one_star_forward_modelling(data, noisemap, psf, subsampling_factor, n_iter=2000,
                                  uniform_background_per_epoch=False, starlet_global_background=True):
    """data, noisemap: (E, n, n) (rescaled IN PLACE by nanmax(data)); psf: (E, N, N) narrow PSFs.

    Returns dict: scale, kwargs_final, fluxes, fluxes_uncertainties, chi2, chi2_per_frame, loss_curve,
    residuals, deconvolved_image, starlet_background.
    """
    scale = np.nanmax(data)
    data /= scale
    noisemap /= scale
    variance = noisemap ** 2
    n_epochs = len(data)

    # rough aperture-like flux guess: stamp sum minus the border sky level
    flux_guess = np.nansum(data, axis=(1, 2)) - data[0].size * _border_level(data)
    model, k_init, k_up, k_down, _ = setup_model(data, variance, psf, np.array([0.]), np.array([0.]),
                                                 subsampling_factor, list(flux_guess))

    # what stays fixed: rotation always; background grid and per-epoch constant unless asked for
    fixed = {'kwargs_analytic': {'alpha': k_init['kwargs_analytic']['alpha']},
             'kwargs_background': {}, 'kwargs_sersic': {}}
    if not starlet_global_background:
        fixed['kwargs_background']['h'] = k_init['kwargs_background']['h']
    if not uniform_background_per_epoch:
        fixed['kwargs_background']['mean'] = np.zeros(n_epochs)
    pars = ParametersDeconv(kwargs_init=k_init, kwargs_fixed=fixed, kwargs_up=k_up, kwargs_down=k_down)

    loss_options = dict(data=data, deconv_class=model, param_class=pars, sigma_2=variance,
                        regularization_terms='l1_starlet', regularization_strength_scales=3.0,
                        regularization_strength_hf=3.0, regularization_strength_flux_uniformity=0.)
    if starlet_global_background:
        loss_options['W'] = propagate_noise(model, noisemap, k_init, wavelet_type_list=['starlet'], method='SLIT',
                                            num_samples=200, seed=1, likelihood_type='chi2', verbose=False,
                                            upsampling_factor=subsampling_factor)[0]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        loss = Loss(**loss_options)
    optim = Optimizer(loss, pars, method='adabelief')
    optim.minimize(max_iterations=n_iter, min_iterations=None, init_learning_rate=1e-3, schedule_learning_rate=True,
                   restart_from_init=True, stop_at_loss_increase=False, progress_bar=True,
                   return_param_history=True)   # as the reference passes it (:119): recorded on the device, copied only if read
    k_final = pars.best_fit_values(as_kwargs=True)

    residuals = data - np.array(model.model(k_final))
    chi2_per_frame = np.nansum(residuals ** 2 / variance, axis=(1, 2)) / model.image_size ** 2
    sigma_a = get_flux_uncertainties(kwargs=k_final, kwargs_down=k_down, kwargs_up=k_up, data=data,
                                     noisemap=noisemap, model=model)
    scene, background = model.getDeconvolved(k_final, 0)
    return {
        'scale': scale,
        'kwargs_final': k_final,
        'fluxes': scale * np.array(k_final['kwargs_analytic']['a']),
        'fluxes_uncertainties': scale * sigma_a,
        'chi2': float(np.nanmean(chi2_per_frame)),
        'chi2_per_frame': np.array(chi2_per_frame),
        'loss_curve': optim.loss_history,
        'residuals': scale * residuals,
        'deconvolved_image': scale * scene,
        'starlet_background': scale * background,
    }


def do_many_stars_forward_modelling(stacks, subsampling_factor, n_iter=2000, uniform_background_per_epoch=False,
                                    starlet_global_background=False):
    """The reference's loop over its reference stars (star_photometry.py:257-326: ``do_one_star_forward_modelling`` once per
    star, each a 2000-iteration fit) as ONE batched device fit (``lightcurver_amd.joint.StarPhotometryBatch``,
    lc_joint_create_groups): every AdaBelief iteration advances all stars with one kernel pair instead of one pair per star.

    stacks: list of (data, noisemap, psf) per star - (E_g, n, n), (E_g, n, n), (E_g, N, N); the epoch counts may differ
    (each star keeps its own frame selection), the stamp size may not.  data and noisemap are rescaled IN PLACE by
    nanmax(data), as the one-star function does.  The two switches are the one-star function's (config.yaml:254-258
    ``star_photometry_uniform_background_per_epoch`` / ``star_photometry_starlet_global_background``; NOTE the defaults here are
    the pipeline's configuration, both off, whereas the one-star function's own default is ``starlet_global_background=True``):

    * both off (the pipeline's default): fluxes, the star's position and the per-epoch shifts free, no background - batched;
    * ``uniform_background_per_epoch=True``: the sky level of every epoch free as well - batched (one more per-epoch
      parameter of the same update);
    * ``starlet_global_background=True``: every star its own background grid with the starlet regulariser.  The batched object
      carries no background work space (no spectra, no slabs: lc_joint_create_groups), so these fits run one star after the
      other through ``do_one_star_forward_modelling`` - same numbers, no batching gain.

    Returns one dictionary per star with ALL keys of ``do_one_star_forward_modelling`` (``deconvolved_image``: the scene of the
    star's first epoch, ``starlet_background``: zeros when no background is fitted - what the reference's caller reads at
    star_photometry.py:139-150); each star's numbers are bit for bit those of its own one-star fit."""
    from ..joint import StarPhotometryBatch, EmbeddedJointFit, joint_fit_size
    from ..starred.deconvolution.deconvolution import nest_kwargs
    ss = int(subsampling_factor)
    if not stacks:
        return []
    shapes = {np.asarray(d).shape[1:] for d, _, _ in stacks}
    n_user = int(np.asarray(stacks[0][0]).shape[-1])
    if shapes != {(n_user, n_user)}:
        raise ValueError(f'every star of a batch needs square stamps of ONE size, got {sorted(shapes)}')
    for d, nm, p in stacks:
        if np.asarray(nm).shape != np.asarray(d).shape or np.asarray(p).shape != (len(d), n_user * ss, n_user * ss):
            raise ValueError('every star needs data, noisemap (E_g, n, n) and psf (E_g, n ss, n ss)')
    if starlet_global_background:
        return [do_one_star_forward_modelling(d, nm, p, ss, n_iter=n_iter, uniform_background_per_epoch=uniform_background_per_epoch,
                                              starlet_global_background=True) for d, nm, p in stacks]
    # a stamp size without a kernel of its own: the stamps in the centre of the next instantiated size, no weight on the ring
    # (what joint.EmbeddedJointFit does for the one-star fit; no background here, so only the model has to be cut back)
    n_fit = joint_fit_size(n_user, ss)
    pad = (n_fit - n_user) // 2

    def embed(a, fill, f=1):
        if pad == 0:
            return a
        a = np.asarray(a)
        out = np.full((a.shape[0], n_fit * f, n_fit * f), fill, dtype=a.dtype)
        out[:, pad * f:pad * f + n_user * f, pad * f:pad * f + n_user * f] = a
        return out

    scales, guesses, dev_stacks = [], [], []
    for data, noisemap, psf in stacks:
        scale = np.nanmax(data)
        data /= scale
        noisemap /= scale
        scales.append(scale)
        guesses.append(np.nansum(data, axis=(1, 2)) - data[0].size * _border_level(data))
        dev_stacks.append((embed(data, 0.0), embed(noisemap ** 2, EmbeddedJointFit.RING_VARIANCE), embed(psf, 0.0, ss)))
    batch = StarPhotometryBatch(dev_stacks, ss, M=1)
    try:
        E, G, N = batch.E, batch.G, batch.N
        batch.set_params(a=np.concatenate(guesses), c_x=np.zeros(G), c_y=np.zeros(G), dx=np.zeros(E), dy=np.zeros(E),
                         alpha=np.zeros(E), h=np.zeros(N * N), mean=np.zeros(E))
        batch.set_loss(lam_scales=3.0, lam_hf=3.0)          # (constants with the background fixed at zero)
        batch.set_free(['a', 'c_x', 'c_y', 'dx', 'dy'] + (['mean'] if uniform_background_per_epoch else []))
        batch.run_adabelief(int(n_iter), init_learning_rate=1e-3, schedule_learning_rate=True)
        final = batch.get_params()
        hist = batch.loss_history()
        model, _ = batch.model()
        model = model[:, pad:pad + n_user, pad:pad + n_user]
        sigma_a = batch.fisher_flux_sigma()
        P, Nu = pad * ss, n_user * ss
        scenes = [batch.deconvolved(int(batch.starts[g]))[0][P:P + Nu, P:P + Nu] for g in range(G)]
    finally:
        batch.close()
    out = []
    for g, (data, noisemap, psf) in enumerate(stacks):
        e0, e1 = batch.starts[g], batch.starts[g + 1]
        flat = dict(a=final['a'][e0:e1], c_x=final['c_x'][g:g + 1], c_y=final['c_y'][g:g + 1], dx=final['dx'][e0:e1],
                    dy=final['dy'][e0:e1], alpha=final['alpha'][e0:e1], h=np.zeros((n_user * ss) ** 2, np.float32), mean=final['mean'][e0:e1])
        k_final = nest_kwargs(flat)
        residuals = data - model[e0:e1]
        chi2_per_frame = np.nansum(residuals ** 2 / noisemap ** 2, axis=(1, 2)) / data.shape[1] ** 2
        scale = scales[g]
        out.append({
            'scale': scale,
            'kwargs_final': k_final,
            'fluxes': scale * np.array(k_final['kwargs_analytic']['a']),
            'fluxes_uncertainties': scale * sigma_a[e0:e1],
            'chi2': float(np.nanmean(chi2_per_frame)),
            'chi2_per_frame': np.array(chi2_per_frame),
            'loss_curve': np.asarray(hist[g, 1:], dtype=np.float64).tolist(),
            'residuals': scale * residuals,
            'deconvolved_image': scale * np.ascontiguousarray(scenes[g]),
            'starlet_background': scale * np.zeros((n_user * ss, n_user * ss), np.float32),
        })
    return out
