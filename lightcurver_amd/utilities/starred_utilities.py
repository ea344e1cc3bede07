"""Flux uncertainties from the diagonal Fisher information, the way the reference's
lightcurver/utilities/starred_utilities.py:10-39 obtains them from STARRED: every parameter except the
fluxes is frozen at its fitted value, the fluxes are polished by a few L-BFGS-B iterations and the
1-sigma is read off the Hessian diagonal of the chi2."""
from copy import deepcopy

import numpy as np

from ..starred.deconvolution.loss import Loss
from ..starred.deconvolution.parameters import ParametersDeconv
from ..starred.optim.inference_base import FisherCovariance
from ..starred.optim.optimization import Optimizer


def get_flux_uncertainties(kwargs, kwargs_up, kwargs_down, data, noisemap, model, refine_iterations=10):
    """One uncertainty per entry of kwargs['kwargs_analytic']['a'] (same epoch-major order)."""
    frozen = deepcopy(kwargs)
    frozen['kwargs_analytic'].pop('a')
    pars = ParametersDeconv(kwargs_init=kwargs, kwargs_fixed=frozen, kwargs_up=kwargs_up, kwargs_down=kwargs_down)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')  # "lambda is not normalized": h is frozen here, the term is a constant
        loss = Loss(data, model, pars, np.asarray(noisemap) ** 2, regularization_terms='l1_starlet')
    optim = Optimizer(loss, pars, method='l-bfgs-b')
    if refine_iterations > 0:
        optim.minimize(maxiter=int(refine_iterations))
    fisher = FisherCovariance(pars, optim, diagonal_only=True)
    fisher.compute_fisher_information()
    return np.array(fisher.get_kwargs_sigma()['kwargs_analytic']['a'])
