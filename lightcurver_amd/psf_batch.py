"""Thin Python object over the lc_psf_batch_* entry points (one batch = F frames x S stars)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import f32, ptr


class PsfBatch:
    """Device-resident batch of PSF fits; mirrors the arithmetic STARRED's build_psf performs per
    frame (reference: lightcurver/processes/psf_modelling.py:164-171)."""

    def __init__(self, data, weight, ss, ctx=None):
        data = f32(data)
        weight = f32(weight)
        if data.ndim != 4 or data.shape != weight.shape or data.shape[2] != data.shape[3]:
            raise ValueError('data and weight must be (F, S, n, n)')
        self.ctx = ctx or _lib.default_context()
        self._l = _lib.lib()
        self.F, self.S, self.n, _ = data.shape
        self.ss = int(ss)
        self.N = self.n * self.ss
        self.J = int(np.log2(self.N))
        if not self._l.lc_psf_supported(self.n, self.ss):
            raise _lib.LcError(f'no PSF kernel for stamp size n={self.n}, subsampling {self.ss}')
        h = C.c_void_p()
        self.ctx.check(self._l.lc_psf_batch_create(self.ctx.h, self.F, self.S, self.n, self.ss, ptr(data),
                                                   ptr(weight), C.byref(h)), 'lc_psf_batch_create')
        self.h = h

    def _chk(self, rc, what):
        self.ctx.check(rc, what)

    def set_moffat(self, moffat):
        m = f32(moffat).reshape(self.F, 4)
        self._chk(self._l.lc_psf_batch_set_moffat(self.h, ptr(m)), 'set_moffat')

    def get_moffat(self):
        m = np.empty((self.F, 4), np.float32)
        self._chk(self._l.lc_psf_batch_get_moffat(self.h, ptr(m)), 'get_moffat')
        return m

    # -- field distortion (include/lcmi.h, "build_psf(field_distortion=True)") -----------------------------------
    def set_moffat_q(self, q):
        """Moffat of every frame by its quadratic form: q [F][4] = q11, q12, q22, beta."""
        q = f32(q).reshape(self.F, 4)
        self._chk(self._l.lc_psf_batch_set_moffat_q(self.h, ptr(q)), 'set_moffat_q')

    def set_distortion(self, S_stars, coeffs, xy):
        c = f32(coeffs).reshape(self.F, 9)
        p = f32(xy).reshape(self.F, int(S_stars), 2)
        self._chk(self._l.lc_psf_batch_set_distortion(self.h, int(S_stars), ptr(c), ptr(p)), 'set_distortion')

    def distortion_forward(self, stars):
        self._chk(self._l.lc_psf_distortion_forward(self.h, stars.h), 'distortion_forward')

    def distortion_backward(self, stars):
        self._chk(self._l.lc_psf_distortion_backward(self.h, stars.h), 'distortion_backward')

    def distortion_run(self, stars, n_iter, **cfg):
        """The whole pixel-grid stage of the distortion fit in one call (lc_psf_distortion_run): n_iter times
        { forward; step of ``stars``; backward; step of this batch } and a final forward, enqueued from C++."""
        c = _lib.adabelief_cfg(**cfg)
        self._chk(self._l.lc_psf_distortion_run(self.h, stars.h, int(n_iter), C.byref(c)), 'distortion_run')

    def get_ext_grad(self):
        g = np.empty((self.F, self.N, self.N), np.float32)
        self._chk(self._l.lc_psf_batch_get_ext_grad(self.h, ptr(g)), 'get_ext_grad')
        return g

    def step_adabelief(self, use_ext_grad=False, export_grad=False, **cfg):
        c = _lib.adabelief_cfg(**cfg)
        self._chk(self._l.lc_psf_batch_step_adabelief(self.h, C.byref(c), int(bool(use_ext_grad)), int(bool(export_grad))),
                  'step_adabelief')

    def set_stars(self, stars):
        s = f32(stars).reshape(self.F, self.S, 4)
        self._chk(self._l.lc_psf_batch_set_stars(self.h, ptr(s)), 'set_stars')

    def get_stars(self):
        s = np.empty((self.F, self.S, 4), np.float32)
        self._chk(self._l.lc_psf_batch_get_stars(self.h, ptr(s)), 'get_stars')
        return s

    def set_grid(self, grid=None):
        g = None if grid is None else f32(grid).reshape(self.F, self.N * self.N)
        self._chk(self._l.lc_psf_batch_set_grid(self.h, ptr(g)), 'set_grid')

    def get_grid(self):
        g = np.empty((self.F, self.N, self.N), np.float32)
        self._chk(self._l.lc_psf_batch_get_grid(self.h, ptr(g)), 'get_grid')
        return g

    def set_regularization(self, W=None, lam_scales=1.0, lam_hf=1.0):
        w = None if W is None else f32(W).reshape(self.F, self.J, self.N, self.N)
        self._chk(self._l.lc_psf_batch_set_regularization(self.h, ptr(w), float(lam_scales), float(lam_hf)),
                  'set_regularization')

    def propagate_noise(self):
        self._chk(self._l.lc_psf_batch_propagate_noise(self.h), 'propagate_noise')

    def get_weights(self):
        w = np.empty((self.F, self.J, self.N, self.N), np.float32)
        self._chk(self._l.lc_psf_batch_get_weights(self.h, ptr(w)), 'get_weights')
        return w

    def evaluate(self, model=False):
        F, S, n, N = self.F, self.S, self.n, self.N
        out = dict(loss=np.empty(F, np.float32), chi2=np.empty(F, np.float32),
                   grad_moffat=np.empty((F, 4), np.float32), grad_stars=np.empty((F, S, 4), np.float32),
                   grad_grid=np.empty((F, N, N), np.float32))
        mod = np.empty((F, S, n, n), np.float32) if model else None
        self._chk(self._l.lc_psf_batch_eval(self.h, ptr(out['loss']), ptr(out['chi2']), ptr(out['grad_moffat']),
                                            ptr(out['grad_stars']), ptr(out['grad_grid']), ptr(mod)), 'eval')
        if model:
            out['model'] = mod
        return out

    def fit_moffat(self, n_iter):
        fl = np.empty(self.F, np.float32)
        self._chk(self._l.lc_psf_batch_fit_moffat(self.h, int(n_iter), ptr(fl)), 'fit_moffat')
        return fl

    def run_adabelief(self, n_iter, **cfg):
        c = _lib.adabelief_cfg(**cfg)
        self._chk(self._l.lc_psf_batch_run_adabelief(self.h, int(n_iter), C.byref(c)), 'run_adabelief')

    @property
    def iterations_done(self):
        return self._l.lc_psf_batch_iterations_done(self.h)

    @property
    def split_fallbacks(self):
        """Two-workgroup launches of this batch that gave up and were redone in the one-workgroup form."""
        c = C.c_int()
        self._chk(self._l.lc_psf_batch_split_fallbacks(self.h, C.byref(c)), 'split_fallbacks')
        return c.value

    def loss_history(self):
        """(F, T + 1): loss at theta_0 .. theta_T."""
        T = self.iterations_done
        h = np.empty((self.F, T + 1), np.float32)
        self._chk(self._l.lc_psf_batch_get_loss_history(self.h, ptr(h), T + 1), 'get_loss_history')
        return h

    def results(self):
        F, S, n, N = self.F, self.S, self.n, self.N
        narrow = np.empty((F, N, N), np.float32)
        full = np.empty((F, N, N), np.float32)
        res = np.empty((F, S, n, n), np.float32)
        chi2 = np.empty(F, np.float32)
        self._chk(self._l.lc_psf_batch_get_results(self.h, ptr(narrow), ptr(full), ptr(res), ptr(chi2)), 'get_results')
        return dict(narrow_psf=narrow, full_psf=full, residuals=res, chi2=chi2)

    def close(self):
        if getattr(self, 'h', None):
            self._l.lc_psf_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        # (not while the interpreter shuts down: objects are then torn down in no particular order, and destroying a device
        #  object whose context has already gone is a crash at exit; the process is about to return everything anyway)
        try:
            import sys
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass
