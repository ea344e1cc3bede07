"""``Loss`` and ``Prior`` of the deconvolution (reference call sites:
lightcurver/processes/star_photometry.py:95-111; roi_modelling.py:240-244,275-276,313-321;
utilities/starred_utilities.py:31-32)."""
import warnings

import numpy as np


class Prior:
    """Gaussian priors ``[[name, mean, sigma], ...]``; this path uses them on c_x / c_y only
    (roi_modelling.py:240-244)."""

    def __init__(self, prior_analytic=None, prior_background=None, prior_sersic=None):
        self.terms = {}
        for name, mean, sigma in (prior_analytic or []):
            if name not in ('c_x', 'c_y'):
                raise NotImplementedError(f'prior on {name!r}: only c_x / c_y priors are built')
            self.terms[name] = (np.atleast_1d(np.asarray(mean, dtype=np.float64)),
                                np.atleast_1d(np.asarray(sigma, dtype=np.float64)))
        if prior_background or prior_sersic:
            raise NotImplementedError('only analytic (c_x, c_y) priors are built')

    def as_arrays(self, M, c_x, c_y):
        """Arrays for the C ABI; a missing axis gets an (effectively) flat prior."""
        big = 1e15
        mx, sx = self.terms.get('c_x', (np.asarray(c_x, dtype=np.float64), np.full(M, big)))
        my, sy = self.terms.get('c_y', (np.asarray(c_y, dtype=np.float64), np.full(M, big)))
        return dict(c_x_mean=np.broadcast_to(mx, (M,)), c_x_sigma=np.broadcast_to(sx, (M,)),
                    c_y_mean=np.broadcast_to(my, (M,)), c_y_sigma=np.broadcast_to(sy, (M,)))


class Loss:
    """0.5 chi2 + l1-starlet(h) + positivity + flux terms + prior, evaluated and differentiated on the GPU."""

    def __init__(self, data, deconv_class, param_class, sigma_2, regularization_terms='l1_starlet',
                 regularization_strength_scales=1.0, regularization_strength_hf=1.0,
                 regularization_strength_positivity=0., regularization_strength_positivity_ps=0.,
                 regularization_strength_pts_source=0., regularization_strength_flux_uniformity=0.,
                 W=None, regularize_full_model=False, prior=None):
        if regularization_terms not in ('l1_starlet', None):
            raise NotImplementedError(f'regularization_terms={regularization_terms!r}')
        if regularize_full_model:
            raise NotImplementedError('regularize_full_model=True')
        if W is None and regularization_terms == 'l1_starlet':
            warnings.warn('lambda is not normalized. Provide the weight map !')
        self._deconv = deconv_class
        self._param = param_class
        self._fit = deconv_class._ensure_fit(data, sigma_2)
        lam_sc = float(regularization_strength_scales) if regularization_terms else 0.0
        lam_hf = float(regularization_strength_hf) if regularization_terms else 0.0
        prior_arrays = None
        if prior is not None:
            cur = param_class._current
            prior_arrays = prior.as_arrays(deconv_class.M, cur['c_x'], cur['c_y'])
        self._settings = dict(W=None if W is None else np.asarray(W), lam_scales=lam_sc, lam_hf=lam_hf,
                              lam_positivity=float(regularization_strength_positivity),
                              lam_positivity_ps=float(regularization_strength_positivity_ps),
                              lam_pts_source=float(regularization_strength_pts_source),
                              lam_flux_uniformity=float(regularization_strength_flux_uniformity),
                              prior=prior_arrays)
        self.data = data
        self.sigma_2 = sigma_2

    def configure(self):
        """Push the loss settings to the device object (called by the optimiser before it runs)."""
        # several Loss objects may share one device fit; the settings (with the weight cube) are uploaded only when
        # another Loss, or a direct set_loss call, configured it last
        if getattr(self._fit, '_configured_by', None) is not self:
            self._fit.set_loss(**self._settings)
            self._fit._configured_by = self
        return self._fit

    def value_and_grad(self, args):
        fit = self.configure()
        flat = self._param.args2flat(args)
        fit.set_params(**flat)
        loss, g = fit.loss_grad(tuple(self._param.free))
        grad = np.concatenate([g[k].astype(np.float64) for k in self._param.free]) if self._param.free else np.zeros(0)
        return float(loss), grad

    def loss(self, args):
        return self.value_and_grad(args)[0]

    __call__ = loss

    def gradient(self, args):
        return self.value_and_grad(args)[1]
