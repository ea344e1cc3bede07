"""``propagate_noise(method='SLIT', likelihood_type='chi2')`` (reference call sites:
lightcurver/processes/star_photometry.py:108-110, roi_modelling.py:299-301)."""
import numpy as np

from ...joint import make_joint_fit


def propagate_noise(model, noise_maps, kwargs=None, masks=None, wavelet_type_list=('starlet',), method='SLIT',
                    num_samples=200, seed=None, likelihood_type='chi2', verbose=False, upsampling_factor=1,
                    scaling_noise_ref=None):
    """Noise level of the chi2 gradient in every starlet scale of the background plane.
    Returns a list with one (J + 1, N, N) array per wavelet type."""
    if list(wavelet_type_list) != ['starlet']:
        raise NotImplementedError("only wavelet_type_list=['starlet']")
    if method not in ('SLIT', 'MC'):
        raise ValueError(f'unknown method {method!r}')
    if likelihood_type != 'chi2':
        raise NotImplementedError("only likelihood_type='chi2'")
    if int(upsampling_factor) != model.upsampling_factor:
        raise ValueError('upsampling_factor differs from the model')
    # 'MC' converges to the analytic 'SLIT' levels; both are served by the deterministic propagation
    noise_maps = np.asarray(noise_maps, dtype=np.float64)
    sigma2 = noise_maps ** 2
    fit = model._fit
    temp = None
    # the model's device object can serve when it was built on these very variances (the fp32 values the device holds)
    same = (fit is not None and fit.E == noise_maps.shape[0] and getattr(model, '_sigma2_f32', None) is not None
            and model._sigma2_f32.shape == sigma2.shape and np.array_equal(model._sigma2_f32, sigma2.astype(np.float32)))
    if not same:
        temp = fit = make_joint_fit(np.zeros_like(sigma2), sigma2, model.psf, model.upsampling_factor, model.M, model._ctx)
    try:
        W = fit.propagate_noise()
    finally:
        if temp is not None:
            temp.close()
    return [W]
