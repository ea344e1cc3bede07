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
        """A history handed out lazily (starred/optim/optimization.py: a weak reference to it is kept here) comes to the host
        before its buffer goes away - if somebody still holds it; a history nobody kept is dropped on the device, uncopied."""
        ref = getattr(self, '_pending_param_history', None)
        if ref is not None:
            self._pending_param_history = None
            pending = ref()
            if pending is not None:
                pending.materialize()
            elif getattr(self, 'h', None):
                self._l.lc_joint_param_history_end(self.h)

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

    def cluster_info(self):
        """(workgroups per epoch of the most recent epoch launch - 0 = the one-workgroup kernel -, runs in which a cluster
        launch gave up and the library fell back): lc_joint_cluster_info."""
        parts, fb = C.c_int(), C.c_int()
        self._chk(self._l.lc_joint_cluster_info(self.h, C.byref(parts), C.byref(fb)), 'cluster_info')
        return parts.value, fb.value

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


def joint_fit_size(n, ss):
    """The stamp size the device fits an n x n joint problem at: n itself when an epoch kernel is instantiated for it,
    otherwise the next instantiated size of the same parity (EmbeddedJointFit)."""
    l = _lib.lib()
    if l.lc_joint_supported(int(n), int(ss)):
        return int(n)
    for m in range(int(n) + 2, 129, 2):
        if l.lc_joint_supported(m, int(ss)):
            return m
    raise _lib.LcError(f'no joint-fit kernel for {n}x{n} stamps at subsampling {ss} (instantiated at ss = 2: multiples of 8 up '
                       f'to 64, and 128; other even sizes up to 128 are fitted embedded in the next of these)')


def make_joint_fit(data, sigma2, psf, ss, M, ctx=None):
    """JointFit at a size with a kernel of its own, EmbeddedJointFit otherwise."""
    n = np.asarray(data).shape[-1]
    n_fit = joint_fit_size(n, ss)
    if n_fit == n:
        return JointFit(data, sigma2, psf, ss, M, ctx)
    return EmbeddedJointFit(data, sigma2, psf, ss, M, ctx, n_fit)


class EmbeddedJointFit(JointFit):
    """The reference's ``stamp_size_ROI`` / ``stamp_size_stars`` are free integers (config.yaml:205-206); the epoch kernels exist
    for n = 8 k <= 64 and 128.  An even n in between is fitted EMBEDDED in the next of those, as a declared fall-back: the
    stamps in the centre of the larger frame, no weight on the ring (variance 1e20), the narrow PSFs zero-padded, the
    background h fitted on the larger grid.  Towards the caller everything keeps the caller's size: ``h`` goes in padded
    with zeros and comes out cropped (values, gradients, bounds, parameter history), models and deconvolved images are
    cropped.  Inside the caller's window the model is the native-size one (the convolution window is alias-free in both)
    except at the edges: a translated epoch samples h with zeros beyond the window where the native grid clamps to its
    edge pixel, and the part of a point source's profile that falls beyond the window is convolved back in here and cut
    there - 1e-4 ... 1e-3 of the model's peak on the synthetic ROIs of tests/test_joint_sizes_gpu.py, growing with the
    background's amplitude at the edge.  (Measured and left out: the ring of h as a copy of the edge - the scene beyond
    the window is then not zero any more and convolves into it: three times further from the native model.)  Also
    different from a fit at a kernel of its own: the regulariser sees the larger grid (J = log2 of it, other edges), and
    the ring pixels of h are free, held at zero by the regulariser alone.  The noise levels W of the regulariser are propagated with the ring at the stamps'
    median variance (a ring of 'infinite' noise would inflate every coarse scale).  Sharded over ranks like a native-size fit
    (tests/test_joint_sizes_gpu.py)."""
    RING_VARIANCE = 1e20

    def __init__(self, data, sigma2, psf, ss, M, ctx, n_fit):
        data, sigma2, psf = f32(data), f32(sigma2), f32(psf)
        E, n, _ = data.shape
        ss = int(ss)
        if (n_fit - n) % 2 or n_fit <= n:
            raise ValueError('embedding needs a larger size of the same parity')
        if psf.shape != (E, n * ss, n * ss):
            raise ValueError(f'psf must be ({E}, {n * ss}, {n * ss}), got {psf.shape}')
        self.pad, self.n_user, self.N_user = (n_fit - n) // 2, n, n * ss
        p, P = self.pad, self.pad * ss
        big = np.zeros((E, n_fit, n_fit), np.float32)
        big[:, p:p + n, p:p + n] = data
        var = np.full((E, n_fit, n_fit), self.RING_VARIANCE, np.float32)
        var[:, p:p + n, p:p + n] = sigma2
        self._psf_fit = np.zeros((E, n_fit * ss, n_fit * ss), np.float32)
        self._psf_fit[:, P:P + n * ss, P:P + n * ss] = psf
        # (for the noise propagation: the ring at each epoch's median variance)
        self._var_w = var.copy()
        med = np.median(sigma2.reshape(E, -1), axis=1).astype(np.float32)
        ring = np.ones((n_fit, n_fit), bool)
        ring[p:p + n, p:p + n] = False
        self._var_w[:, ring] = med[:, None]
        super().__init__(big, var, self._psf_fit, ss, M, ctx)
        self._dev_dims = (self.n, self.N, dict(self.sizes))
        self._user_dims = (n, n * ss, dict(self.sizes, h=(n * ss) ** 2))
        self.n, self.N, self.sizes = self._user_dims

    class _Dev:
        def __init__(self, fit):
            self.fit = fit

        def __enter__(self):
            self.fit.n, self.fit.N, self.fit.sizes = self.fit._dev_dims

        def __exit__(self, *exc):
            self.fit.n, self.fit.N, self.fit.sizes = self.fit._user_dims

    def _dev(self):
        return EmbeddedJointFit._Dev(self)

    def _crop_h(self, v):
        Nf, Nu, P = self._dev_dims[1], self.N_user, self.pad * self.ss
        return np.ascontiguousarray(np.asarray(v).reshape(Nf, Nf)[P:P + Nu, P:P + Nu]).ravel()

    def _pad_h(self, v, fill=0.0):
        Nf, Nu, P = self._dev_dims[1], self.N_user, self.pad * self.ss
        out = np.full((Nf, Nf), fill, np.float32)
        out[P:P + Nu, P:P + Nu] = np.broadcast_to(np.asarray(v, dtype=np.float32).ravel(), (Nu * Nu,)).reshape(Nu, Nu)
        return out.ravel()

    def set_params(self, **params):
        if 'h' in params:
            params = dict(params, h=self._pad_h(params['h']))
        with self._dev():
            super().set_params(**params)

    def get_params(self, names=None):
        names = list(names or self.sizes)
        with self._dev():
            out = super().get_params(names)
        if 'h' in out:
            out['h'] = self._crop_h(out['h'])
        return out

    def set_loss(self, **kw):
        with self._dev():
            super().set_loss(**kw)

    def propagate_noise(self):
        with self._dev():
            data0 = np.zeros((self.E, self.n, self.n), np.float32)
            temp = JointFit(data0, self._var_w, self._psf_fit, self.ss, self.M, self.ctx)
        try:
            return temp.propagate_noise()      # (J + 1, N_fit, N_fit): handed back to set_loss as it is
        finally:
            temp.close()

    def loss_grad(self, names=('a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean')):
        with self._dev():
            loss, bufs = super().loss_grad(names)
        if 'h' in bufs:
            bufs['h'] = self._crop_h(bufs['h'])
        return loss, bufs

    def model(self):
        with self._dev():
            m, chi2 = super().model()
        p, n = self.pad, self.n_user
        return np.ascontiguousarray(m[:, p:p + n, p:p + n]), chi2

    def deconvolved(self, epoch=0):
        with self._dev():
            s, b = super().deconvolved(epoch)
        P, Nu = self.pad * self.ss, self.N_user
        return np.ascontiguousarray(s[P:P + Nu, P:P + Nu]), np.ascontiguousarray(b[P:P + Nu, P:P + Nu])

    def run_lbfgs(self, maxiter, lower=None, upper=None):
        def ring(d, fill):
            return None if d is None else {k: (self._pad_h(v, fill) if k == 'h' else v) for k, v in d.items()}
        with self._dev():
            return super().run_lbfgs(maxiter, ring(lower, -1e10), ring(upper, 1e10))

    def param_history_begin(self, capacity):
        super().param_history_begin(capacity)
        return sum(self.sizes[k] for k in self._free_names)

    def param_history(self, first=0, count=None):
        with self._dev():
            rows = super().param_history(first, count)
            dev_sizes = self.sizes
        if 'h' not in self._free_names:
            return rows
        off = 0
        parts = []
        for k in self._free_names:
            blk = rows[:, off:off + dev_sizes[k]]
            off += dev_sizes[k]
            if k == 'h':
                Nf, Nu, P = self._dev_dims[1], self.N_user, self.pad * self.ss
                blk = blk.reshape(-1, Nf, Nf)[:, P:P + Nu, P:P + Nu].reshape(rows.shape[0], Nu * Nu)
            parts.append(blk)
        return np.ascontiguousarray(np.concatenate(parts, axis=1))

    # Sharded over ranks (round 4): every rank embeds its epochs in the same larger frame, so the shared block - the gradient of
    # h on the LARGER grid, the sums over the sources - means the same on all of them and step_local / the all-reduce /
    # step_update / run_sharded are the native-size calls unchanged; only what crosses to the caller changes size.
    def step_grad(self, names=('a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean')):
        with self._dev():
            loss, bufs = super().step_grad(names)
        if 'h' in bufs:
            bufs['h'] = self._crop_h(bufs['h'])
        return loss, bufs
