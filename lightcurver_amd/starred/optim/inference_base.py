"""``FisherCovariance(diagonal_only=True)`` for the one use the reference makes of it: 1-sigma of the
fluxes with everything else fixed (lightcurver/utilities/starred_utilities.py:36-38)."""
import numpy as np

from ..deconvolution.deconvolution import nest_kwargs


class FisherCovariance:
    def __init__(self, param_class, optimizer_class, diagonal_only=True):
        if not diagonal_only:
            raise NotImplementedError('full Fisher matrix: only diagonal_only=True is built')
        if list(param_class.free) != ['a']:
            raise NotImplementedError("Fisher diagonal is built for the fluxes 'a' only (the reference's use)")
        self._param = param_class
        self._optim = optimizer_class
        self._sigma = None

    def compute_fisher_information(self):
        fit = self._optim._loss.configure()
        fit.set_params(**self._param._current)
        self._sigma = fit.fisher_flux_sigma()

    def get_kwargs_sigma(self):
        if self._sigma is None:
            self.compute_fisher_information()
        flat = {k: np.zeros_like(v) for k, v in self._param._current.items()}
        flat['a'] = self._sigma
        return nest_kwargs(flat)
