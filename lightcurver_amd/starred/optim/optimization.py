"""``Optimizer``: AdaBelief with the loop, state and loss history on the device, or scipy's L-BFGS-B on
the host driving device loss/gradient evaluations, exactly the split the reference has (STARRED
Optimizer; call sites lightcurver/processes/star_photometry.py:113-122, roi_modelling.py:278-280,326-334,
utilities/starred_utilities.py:33-34)."""
import time

import numpy as np
from scipy.optimize import minimize as _scipy_minimize


class Optimizer:
    def __init__(self, loss_class, param_class, method='adabelief'):
        if method not in ('adabelief', 'l-bfgs-b'):
            raise NotImplementedError(f"method {method!r}: 'adabelief' and 'l-bfgs-b' are built")
        self._loss = loss_class
        self._param = param_class
        self.method = method
        self.loss_history = []

    def minimize(self, **kwargs):
        t0 = time.time()
        if self.method == 'adabelief':
            out = self._run_adabelief(**kwargs)
        else:
            out = self._run_lbfgsb(**kwargs)
        best_fit, logL, extra = out
        return best_fit, logL, extra, time.time() - t0

    # -- AdaBelief: everything on the device ------------------------------------------------------------
    def _run_adabelief(self, max_iterations=100, min_iterations=None, init_learning_rate=1e-2,
                       schedule_learning_rate=True, restart_from_init=False, stop_at_loss_increase=False,
                       progress_bar=False, return_param_history=False, decay_rate=0.99, transition_steps=10):
        p = self._param
        start = p._start if restart_from_init else p._current
        fit = self._loss.configure()
        fit.set_params(**start)
        fit.set_free(p.free)
        n_iter = int(max_iterations)
        cfg = dict(init_learning_rate=init_learning_rate, schedule_learning_rate=bool(schedule_learning_rate),
                   decay_rate=decay_rate, transition_steps=transition_steps)
        param_history = []
        on_device = n_iter > 0 and not stop_at_loss_increase
        if on_device and return_param_history:
            # return_param_history=True - what the reference's own call sites pass (star_photometry.py:119,
            # roi_modelling.py:331) - records the free blocks after every update in a device-resident history; it comes to
            # the host only if extra_fields['param_history'] is looked at.  A history that does not fit in device memory
            # (n_iter x P floats) must not fail the fit: the rows are then collected by the host loop below
            from ..._lib import LcError
            try:
                rows = fit.param_history_begin(n_iter)
            except LcError:
                on_device = False
            else:
                if rows != p.num_parameters:
                    raise RuntimeError('parameter history: row length differs from the number of free parameters')
        if on_device:
            # one call: the whole loop stays on the device
            fit.run_adabelief(n_iter, **cfg)
            if return_param_history:
                param_history = _DeviceParamHistory(fit, n_iter)
        elif n_iter > 0:
            # the parameter vector after every update (return_param_history) and / or the early stop need the host in
            # the loop: one iteration per call, so that the fit stops AT the update that raised the loss, as the host loop
            # this replaces does; the optimiser state and the iteration count live on the device, so chunking does not
            # change the trajectory
            chunk = 1
            min_it = 0 if min_iterations is None else int(min_iterations)
            done = 0
            while done < n_iter:
                k = min(chunk, n_iter - done)
                fit.run_adabelief(k, **cfg)
                if return_param_history:
                    flat_now = {kk: np.asarray(v, dtype=np.float64) for kk, v in fit.get_params().items()}
                    param_history.append(p.kwargs2args(_nest(flat_now)))
                if stop_at_loss_increase:
                    h = np.asarray(fit.loss_history(), dtype=np.float64)   # loss at theta_0 .. theta_(done + k)
                    t = np.arange(max(done, 1), done + k + 1)
                    if np.any((h[t] > h[t - 1]) & (t >= max(min_it, 1))):    # an update made the loss go up
                        done += k
                        break
                done += k
        hist = fit.loss_history()
        final = fit.get_params()
        flat = {k: np.asarray(v, dtype=np.float64) for k, v in final.items()}
        p.set_best_fit(flat)
        # loss_history[t] = loss after update t (len == the number of iterations run: max_iterations without early stop)
        self.loss_history = np.asarray(hist[1:], dtype=np.float64).tolist()
        extra = {'loss_history': np.array(self.loss_history), 'initial_loss': float(hist[0])}
        if return_param_history:
            extra['param_history'] = param_history
        return p.best_fit_values(), -float(hist[-1]), extra

    # -- L-BFGS-B ----------------------------------------------------------------------------------------
    # default: the library's bounded L-BFGS with parameters, gradients, direction and history on the device
    # (lc_joint_run_lbfgs); LCMI_LBFGS_SCIPY=1: scipy's L-BFGS-B on the host driving device loss / gradient evaluations,
    # the split the reference has, kept as a cross-check.  Both reach the same optimum; the iterates differ.
    def _run_lbfgsb(self, maxiter=100, restart_from_init=False, **_ignored):
        import os
        p = self._param
        start = p._start if restart_from_init else p._current
        fit = self._loss.configure()
        fit.set_params(**start)
        fit.set_free(p.free)
        x0 = p.kwargs2args(_nest(start))
        lo, hi = p.bounds()
        if x0.size and not os.environ.get('LCMI_LBFGS_SCIPY'):
            lower = {k: p._down[k] for k in p.free if k in p._down}
            upper = {k: p._up[k] for k in p.free if k in p._up}
            hist, nit, nev = fit.run_lbfgs(int(maxiter), lower, upper)
            flat = {k: np.asarray(v, dtype=np.float64) for k, v in fit.get_params().items()}
            p.set_best_fit(flat)
            self.loss_history = np.asarray(hist, dtype=np.float64).tolist()
            return p.best_fit_values(), -float(hist[-1]), {'loss_history': np.array(self.loss_history),
                                                           'iterations': nit, 'evaluations': nev}
        hist, last = [], {}

        def fun(x):
            val, grad = self._loss.value_and_grad(x)
            last['x'], last['val'] = np.array(x, copy=True), val   # scipy's callback gets the accepted point: its loss
            return val, grad                                       # was the most recent evaluation, no second one needed

        def record(xk):
            hist.append(last['val'] if 'x' in last and np.array_equal(last['x'], xk) else fun(xk)[0])

        if x0.size == 0:
            val, _ = fun(x0)
            self.loss_history = [val]
            return x0, -val, {'loss_history': np.array([val])}
        res = _scipy_minimize(fun, x0, jac=True, method='L-BFGS-B', bounds=list(zip(lo, hi)),
                              options={'maxiter': int(maxiter)}, callback=record)
        flat = p.args2flat(res.x)
        fit.set_params(**flat)
        p.set_best_fit(flat)
        self.loss_history = hist if hist else [float(res.fun)]
        return res.x, -float(res.fun), {'loss_history': np.array(self.loss_history), 'scipy_result': res}


class _DeviceParamHistory:
    """``extra_fields['param_history']``: the parameter vector after every update (a sequence of ``max_iterations``
    float64 vectors in kwargs2args order).  The rows live on the device until the sequence is first read; then one copy
    brings them over and the device buffer is released.  The fit keeps only a WEAK reference to this object: when the fit
    next needs its history buffer (set_free, another param_history_begin, close) it calls ``materialize()`` if somebody
    still holds the sequence, and just releases the device rows if nobody does (no copy for a history nobody read)."""

    def __init__(self, fit, n_rows):
        import weakref
        self._fit, self._n, self._rows = fit, int(n_rows), None
        fit._pending_param_history = weakref.ref(self)

    def materialize(self):
        if self._rows is None:
            self._rows = np.asarray(self._fit.param_history(0, self._n), dtype=np.float64)
            self._fit.param_history_end()
            ref = getattr(self._fit, '_pending_param_history', None)
            if ref is not None and ref() is self:
                self._fit._pending_param_history = None
            self._fit = None
        return self._rows

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self.materialize()[i]

    def __iter__(self):
        return iter(self.materialize())

    def __array__(self, dtype=None, copy=None):
        a = self.materialize()
        return a if dtype is None else a.astype(dtype)


def _nest(flat):
    from ..deconvolution.deconvolution import nest_kwargs
    return nest_kwargs(flat)
