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
        self._flush_pending_history()
        order = ('a', 'c_x', 'c_y', 'dx', 'dy', 'alpha', 'h', 'mean')
        self._free_names = [name for name in order if name in free]
        mask = (C.c_int32 * P_COUNT)(*[1 if name in free else 0 for name in order])
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

    def step_grad(self, names=('a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean')):
        """Behind step_local() and the all-reduce of the shared block: loss of the whole fit and gradients (shared parameters
        complete, per-epoch parameters for the local epochs), nothing stepped - lc_joint_step_grad."""
        loss = C.c_float()
        bufs = {k: np.empty(self.sizes[k], np.float32) for k in names}
        arr = (_lib.fp * P_COUNT)()
        for k, b in bufs.items():
            arr[PARAM_INDEX[k]] = ptr(b)
        self._chk(self._l.lc_joint_step_grad(self.h, C.byref(loss), arr), 'step_grad')
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

    # return_param_history=True of the reference's call sites: rows of the free blocks, kept on the device
    def param_history_begin(self, capacity):
        """Record the free parameter blocks after every AdaBelief update into a device-resident history of
        ``capacity`` rows (lc_joint_param_history_begin); returns the row length."""
        self._flush_pending_history()
        n = C.c_int()
        self._chk(self._l.lc_joint_param_history_begin(self.h, int(capacity), C.byref(n)), 'param_history_begin')
        return n.value

    def _flush_pending_history(self):
        """A history handed out lazily (starred/optim/optimization.py) comes to the host before its buffer goes away."""
        pending = getattr(self, '_pending_param_history', None)
        if pending is not None:
            self._pending_param_history = None
            pending.materialize()

    def param_history(self, first=0, count=None):
        rows = self._l.lc_joint_param_history_rows(self.h)
        count = rows - first if count is None else int(count)
        # (row length: the free blocks as set_free left them)
        P = sum(self.sizes[k] for k in self._free_names) if getattr(self, '_free_names', None) is not None else None
        if P is None:
            raise _lib.LcError('param_history: set_free was not called through this object')
        out = np.empty((max(count, 0), P), np.float32)
        self._chk(self._l.lc_joint_param_history_get(self.h, int(first), int(count), ptr(out)), 'param_history_get')
        return out

    def param_history_end(self):
        self._chk(self._l.lc_joint_param_history_end(self.h), 'param_history_end')

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

    def run_sharded(self, n_iter, allreduce, user=None, **cfg):
        """n_iter iterations of { step_local; allreduce(user, block, count, stream); step_update } enqueued from C++
        (lc_joint_run_sharded).  allreduce: a ctypes function of type _lib.ALLREDUCE_FN or a raw function pointer."""
        c = _lib.adabelief_cfg(**cfg)
        fn = allreduce if isinstance(allreduce, C.c_void_p) else C.cast(allreduce, C.c_void_p)
        self._chk(self._l.lc_joint_run_sharded(self.h, int(n_iter), C.byref(c), fn, user), 'run_sharded')

    def step_update(self, **cfg):
        c = _lib.adabelief_cfg(**cfg)
        self._chk(self._l.lc_joint_step_update(self.h, C.byref(c)), 'step_update')

    def close(self):
        if getattr(self, 'h', None):
            try:
                self._flush_pending_history()
            except Exception:
                pass
            self._l.lc_joint_destroy(self.h)
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


class StarPhotometryBatch(JointFit):
    """G independent point-source fits (the reference's loop over its reference stars, star_photometry.py:257) as ONE
    device object (lc_joint_create_groups): ``stacks`` is a list of (data, sigma2, psf) per star, each over that star's
    own epochs ((E_g, n, n), (E_g, n, n), (E_g, N, N)); M point sources per star (1 in the reference).  Parameter arrays
    are the per-star arrays concatenated: a (sum E_g * M), c_x / c_y (G * M), dx / dy / mean (sum E_g).  Every AdaBelief
    iteration is one kernel pair for all stars; each star's trajectory is bit for bit that of its own JointFit."""

    def __init__(self, stacks, ss, M=1, ctx=None):
        datas = [f32(d) for d, _, _ in stacks]
        sig2s = [f32(s2) for _, s2, _ in stacks]
        psfs = [f32(p) for _, _, p in stacks]
        if not datas:
            raise ValueError('at least one star')
        self.G = len(datas)
        self.epochs = np.array([d.shape[0] for d in datas], dtype=np.int32)
        self.starts = np.concatenate([[0], np.cumsum(self.epochs)]).astype(int)
        self.n = datas[0].shape[-1]
        self.ss, self.M = int(ss), int(M)
        self.N = self.n * self.ss
        for d, s2, p in zip(datas, sig2s, psfs):
            if d.shape[1:] != (self.n, self.n) or s2.shape != d.shape or p.shape != (d.shape[0], self.N, self.N):
                raise ValueError('every star needs data, sigma2 (E_g, n, n) and psf (E_g, N, N) of one stamp size')
        data, sig2, psf = (np.ascontiguousarray(np.concatenate(a)) for a in (datas, sig2s, psfs))
        self.E = int(self.epochs.sum())
        self.J = int(np.log2(self.N))
        self.ctx = ctx or _lib.default_context()
        self._l = _lib.lib()
        h = C.c_void_p()
        self.ctx.check(self._l.lc_joint_create_groups(self.ctx.h, self.G, self.epochs.ctypes.data_as(_lib.ip), self.M, self.n,
                                                      self.ss, ptr(data), ptr(sig2), ptr(psf), C.byref(h)), 'lc_joint_create_groups')
        self.h = h
        self.sizes = {'a': self.E * self.M, 'c_x': self.G * self.M, 'c_y': self.G * self.M, 'dx': self.E, 'dy': self.E,
                      'alpha': self.E, 'h': self.N * self.N, 'mean': self.E}
        self._keep = None

    def loss_history(self):
        """(G, T + 1): per star, the loss before every update and the loss of the final parameters."""
        T = self.iterations_done
        hist = np.empty((self.G, T + 1), np.float32)
        self._chk(self._l.lc_joint_get_group_loss_history(self.h, ptr(hist), T + 1), 'get_group_loss_history')
        return hist

    def split(self, flat, name):
        """The concatenated block ``name`` as a list of per-star arrays."""
        flat = np.asarray(flat)
        if name in ('c_x', 'c_y'):
            return [flat[g * self.M:(g + 1) * self.M] for g in range(self.G)]
        k = self.M if name == 'a' else 1
        return [flat[self.starts[g] * k:self.starts[g + 1] * k] for g in range(self.G)]
