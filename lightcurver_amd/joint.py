"""Thin Python object over the lc_joint_* entry points: the device-resident joint multi-epoch
forward model (STARRED ``Deconv`` + ``Loss`` + optimiser state)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PARAM_INDEX, P_COUNT, f32, ptr


class JointFit:
    """data, sigma2: (E, n, n); psf: (E, N, N) narrow PSFs, N = ss * n; M point sources."""

    def __init__(self, data, sigma2, psf, ss, M, ctx=None):
        data, sigma2, psf = f32(data), f32(sigma2), f32(psf)
        if data.ndim != 3 or data.shape != sigma2.shape or data.shape[1] != data.shape[2]:
            raise ValueError('data and sigma2 must be (E, n, n)')
        self.E, self.n, _ = data.shape
        self.ss = int(ss)
        self.N = self.n * self.ss
        self.M = int(M)
        if psf.shape != (self.E, self.N, self.N):
            raise ValueError(f'psf must be ({self.E}, {self.N}, {self.N}), got {psf.shape}')
        self.J = int(np.log2(self.N))
        self.ctx = ctx or _lib.default_context()
        self._l = _lib.lib()
        if not self._l.lc_joint_supported(self.n, self.ss):
            raise _lib.LcError(f'no joint-fit kernel for stamp size n={self.n}, subsampling {self.ss}')
        h = C.c_void_p()
        self.ctx.check(self._l.lc_joint_create(self.ctx.h, self.E, self.M, self.n, self.ss, ptr(data), ptr(sigma2),
                                               ptr(psf), C.byref(h)), 'lc_joint_create')
        self.h = h
        self.sizes = {'a': self.E * self.M, 'c_x': self.M, 'c_y': self.M, 'dx': self.E, 'dy': self.E,
                      'alpha': self.E, 'h': self.N * self.N, 'mean': self.E}
        self._keep = None

    def _chk(self, rc, what):
        self.ctx.check(rc, what)

    def set_params(self, **params):
        for k, v in params.items():
            arr = f32(np.ravel(v))
            self._chk(self._l.lc_joint_set_param(self.h, PARAM_INDEX[k], ptr(arr), arr.size), f'set_param({k})')

    def get_params(self, names=None):
        out = {}
        for k in (names or self.sizes):
            arr = np.empty(self.sizes[k], np.float32)
            self._chk(self._l.lc_joint_get_param(self.h, PARAM_INDEX[k], ptr(arr), arr.size), f'get_param({k})')
            out[k] = arr
        return out

    def set_flux_reference(self, ref):
        """Reference flux per source the flux moments of the shared block are centred on (set_params(a=...) sets the
        local mean; every rank of a sharded fit must use the same values)."""
        r = f32(np.ravel(ref))
        self._chk(self._l.lc_joint_set_flux_reference(self.h, ptr(r), r.size), 'set_flux_reference')

    def get_flux_reference(self):
        r = np.empty(self.M, np.float32)
        self._chk(self._l.lc_joint_get_flux_reference(self.h, ptr(r), r.size), 'get_flux_reference')
        return r

    def set_free(self, free):
        mask = (C.c_int32 * P_COUNT)(*[1 if name in free else 0 for name in
                                       ('a', 'c_x', 'c_y', 'dx', 'dy', 'alpha', 'h', 'mean')])
        self._chk(self._l.lc_joint_set_free(self.h, mask), 'set_free')

    def set_loss(self, W=None, lam_scales=0.0, lam_hf=0.0, lam_positivity=0.0, lam_positivity_ps=0.0,
                 lam_pts_source=0.0, lam_flux_uniformity=0.0, prior=None):
        """prior: dict with c_x_mean, c_x_sigma, c_y_mean, c_y_sigma (length M) or None."""
        self._configured_by = None  # see starred/deconvolution/loss.py Loss.configure
        cfg = _lib.JointLossCfg(float(lam_scales), float(lam_hf), float(lam_positivity), float(lam_positivity_ps),
                                float(lam_pts_source), float(lam_flux_uniformity), 0, None, None, None, None)
        keep = []
        if prior is not None:
            arrs = [f32(np.ravel(prior[k])) for k in ('c_x_mean', 'c_x_sigma', 'c_y_mean', 'c_y_sigma')]
            keep = arrs
            cfg.n_prior = self.M
            cfg.prior_cx_mean, cfg.prior_cx_sigma, cfg.prior_cy_mean, cfg.prior_cy_sigma = [ptr(a) for a in arrs]
        w = None
        if W is not None:
            w = f32(np.asarray(W)[:self.J].reshape(self.J, self.N, self.N))
        self._keep = (keep, w)
        self._chk(self._l.lc_joint_set_loss(self.h, C.byref(cfg), ptr(w)), 'set_loss')

    def propagate_noise(self):
        W = np.empty((self.J + 1, self.N, self.N), np.float32)
        self._chk(self._l.lc_joint_propagate_noise(self.h, ptr(W)), 'propagate_noise')
        return W

    def loss_grad(self, names=('a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean')):
        loss = C.c_float()
        bufs = {k: np.empty(self.sizes[k], np.float32) for k in names}
        arr = (_lib.fp * P_COUNT)()
        for k, b in bufs.items():
            arr[PARAM_INDEX[k]] = ptr(b)
        self._chk(self._l.lc_joint_loss_grad(self.h, C.byref(loss), arr), 'loss_grad')
        return loss.value, bufs

    def model(self):
        m = np.empty((self.E, self.n, self.n), np.float32)
        chi2 = np.empty(self.E, np.float32)
        self._chk(self._l.lc_joint_model(self.h, ptr(m), ptr(chi2)), 'model')
        return m, chi2

    def deconvolved(self, epoch=0):
        s = np.empty((self.N, self.N), np.float32)
        b = np.empty((self.N, self.N), np.float32)
        self._chk(self._l.lc_joint_deconvolved(self.h, int(epoch), ptr(s), ptr(b)), 'deconvolved')
        return s, b

    def run_adabelief(self, n_iter, **cfg):
        c = _lib.adabelief_cfg(**cfg)
        self._chk(self._l.lc_joint_run_adabelief(self.h, int(n_iter), C.byref(c)), 'run_adabelief')

    def run_lbfgs(self, maxiter, lower=None, upper=None):
        """Bounded L-BFGS on the free blocks, vectors on the device (lc_joint_run_lbfgs).  lower / upper: dicts
        name -> array (missing = unbounded).  Returns (loss history per accepted iteration, iterations, evaluations)."""
        keep = []
        def table(d):
            arr = (_lib.fp * P_COUNT)()
            for k, v in (d or {}).items():
                a = f32(np.broadcast_to(np.asarray(v, dtype=np.float64), (self.sizes[k],)))
                keep.append(a)
                arr[PARAM_INDEX[k]] = ptr(a)
            return arr
        lo, hi = table(lower), table(upper)
        cap = max(int(maxiter), 1) + 1
        hist = np.zeros(cap, np.float32)
        nit, nev = C.c_int(), C.c_int()
        self._chk(self._l.lc_joint_run_lbfgs(self.h, int(maxiter), lo, hi, ptr(hist), cap, C.byref(nit), C.byref(nev)), 'run_lbfgs')
        return hist[:max(nit.value, 1)].copy(), nit.value, nev.value

    @property
    def iterations_done(self):
        return self._l.lc_joint_iterations_done(self.h)

    def loss_history(self):
        T = self.iterations_done
        hist = np.empty(T + 1, np.float32)
        self._chk(self._l.lc_joint_get_loss_history(self.h, ptr(hist), T + 1), 'get_loss_history')
        return hist

    def fisher_flux_sigma(self):
        s = np.empty(self.E * self.M, np.float32)
        self._chk(self._l.lc_joint_fisher_flux_sigma(self.h, ptr(s)), 'fisher_flux_sigma')
        return s

    # multi-GPU split step (epoch sharding)
    def step_local(self):
        self._chk(self._l.lc_joint_step_local(self.h), 'step_local')

    def shared_buffer(self):
        p = C.c_void_p()
        n = C.c_int()
        self._chk(self._l.lc_joint_shared_buffer_dev(self.h, C.byref(p), C.byref(n)), 'shared_buffer_dev')
        return p.value, n.value

    def shared_get(self):
        _, n = self.shared_buffer()
        buf = np.empty(n, np.float32)
        self._chk(self._l.lc_joint_shared_get(self.h, ptr(buf), n), 'shared_get')
        return buf

    def shared_set(self, buf):
        buf = f32(buf)
        self._chk(self._l.lc_joint_shared_set(self.h, ptr(buf), buf.size), 'shared_set')

    def step_update(self, **cfg):
        c = _lib.adabelief_cfg(**cfg)
        self._chk(self._l.lc_joint_step_update(self.h, C.byref(c)), 'step_update')

    def close(self):
        if getattr(self, 'h', None):
            self._l.lc_joint_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
