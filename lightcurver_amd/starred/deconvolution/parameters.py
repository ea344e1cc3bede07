"""``ParametersDeconv``: free / fixed bookkeeping of the Deconv kwargs (reference call sites:
lightcurver/processes/star_photometry.py:89-92,123; roi_modelling.py:264-267,303-306;
utilities/starred_utilities.py:27-30)."""
from copy import deepcopy

import numpy as np

from .deconvolution import ANALYTIC, BACKGROUND, flatten_kwargs, nest_kwargs

ORDER = ANALYTIC + BACKGROUND


class ParametersDeconv:
    def __init__(self, kwargs_init, kwargs_fixed, kwargs_up=None, kwargs_down=None):
        self._kwargs_init = deepcopy(kwargs_init)
        self._init = flatten_kwargs(kwargs_init)
        self._fixed = flatten_kwargs(kwargs_fixed)
        self._up = flatten_kwargs(kwargs_up) if kwargs_up is not None else {}
        self._down = flatten_kwargs(kwargs_down) if kwargs_down is not None else {}
        missing = [k for k in ORDER if k not in self._init]
        if missing:
            raise KeyError(f'kwargs_init misses {missing}')
        self.free = [k for k in ORDER if k not in self._fixed]
        if 'alpha' in self.free:
            raise NotImplementedError("'alpha' is never optimised on this path (roi_modelling.py:221-222)")
        # current values: fixed entries take the fixed value
        self._current = {k: np.array(self._fixed.get(k, self._init[k]), dtype=np.float64) for k in ORDER}
        self._start = {k: v.copy() for k, v in self._current.items()}
        self._best = None

    # -- vector <-> kwargs ---------------------------------------------------------------------------
    def kwargs2args(self, kwargs):
        flat = flatten_kwargs(kwargs)
        return np.concatenate([flat[k] for k in self.free]) if self.free else np.zeros(0)

    def args2flat(self, args):
        flat = {k: v.copy() for k, v in self._current.items()}
        o = 0
        for k in self.free:
            sz = flat[k].size
            flat[k] = np.asarray(args[o:o + sz], dtype=np.float64)
            o += sz
        return flat

    def args2kwargs(self, args):
        return nest_kwargs(self.args2flat(args))

    def bounds(self):
        lo, hi = [], []
        for k in self.free:
            sz = self._current[k].size
            lo.append(np.broadcast_to(self._down.get(k, -np.inf), (sz,)) if k in self._down else np.full(sz, -np.inf))
            hi.append(np.broadcast_to(self._up.get(k, np.inf), (sz,)) if k in self._up else np.full(sz, np.inf))
        if not lo:
            return np.zeros(0), np.zeros(0)
        return np.concatenate(lo).astype(np.float64), np.concatenate(hi).astype(np.float64)

    # -- values ----------------------------------------------------------------------------------------
    def initial_values(self, as_kwargs=False):
        return nest_kwargs(self._start) if as_kwargs else self.kwargs2args(nest_kwargs(self._start))

    def current_values(self, as_kwargs=False):
        return nest_kwargs(self._current) if as_kwargs else self.kwargs2args(nest_kwargs(self._current))

    def best_fit_values(self, as_kwargs=False):
        flat = self._best if self._best is not None else self._current
        return nest_kwargs(flat) if as_kwargs else self.kwargs2args(nest_kwargs(flat))

    def set_best_fit(self, flat):
        self._best = {k: np.array(v, dtype=np.float64) for k, v in flat.items()}
        self._current = {k: v.copy() for k, v in self._best.items()}

    @property
    def num_parameters(self):
        return int(sum(self._current[k].size for k in self.free))
