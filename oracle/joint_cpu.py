"""ctypes front of oracle/joint_cpu.c (C / OpenMP restatement of the joint fit WITH the pixelated background: ROI modelling).

TEST / MEASUREMENT INFRASTRUCTURE ONLY - see ``oracle/__init__.py``: bench.py's ``cpu_baseline`` leg of the joint-fit
entries times it (kind "port"), tests/ use it as a further checker.  Nothing under ``lightcurver_amd/`` imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {False: os.path.join(_HERE, '_build', 'libjointcpu.so'), True: os.path.join(_HERE, '_build', 'libjointcpu_f64.so')}
_handles = {}
FREE_ORDER = ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h')


def build(double=False, native=False, out=None):
    """gcc -O3 -fopenmp oracle/joint_cpu.c -> oracle/_build/libjointcpu[_f64].so (x86-64-v3: the file built in the build
    container also runs on the GPU box's host CPU)."""
    out = out or _LIBS[bool(double)]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    arch = 'native' if native else 'x86-64-v3'
    subprocess.run(['gcc', '-O3', f'-march={arch}', '-fopenmp', '-fPIC', '-std=c11', '-shared'] +
                   (['-DJC_CPU_DOUBLE'] if double else []) + [os.path.join(_HERE, 'joint_cpu.c'), '-o', out, '-lm'], check=True)
    return out


def lib(double=False, path=None):
    double = bool(double)
    key = (double, path)
    if key not in _handles:
        p = path or _LIBS[double]
        if not os.path.exists(p):
            build(double)
        h = C.CDLL(p)
        ptr = C.POINTER(C.c_double if double else C.c_float)
        scal = C.c_double if double else C.c_float
        dp, vp = C.POINTER(C.c_double), C.c_void_p
        h.jc_cpu_create.restype = vp
        h.jc_cpu_create.argtypes = [C.c_int] * 4 + [ptr] * 3 + [C.c_int]
        h.jc_cpu_destroy.restype = None
        h.jc_cpu_destroy.argtypes = [vp]
        h.jc_cpu_eval.restype = C.c_int
        h.jc_cpu_eval.argtypes = [vp, vp, ptr] + [ptr] * 8 + [dp] + [ptr] * 8 + [C.c_int]
        h.jc_cpu_run.restype = C.c_int
        h.jc_cpu_run.argtypes = [vp, vp, ptr] + [ptr] * 8 + [ptr, ptr, C.POINTER(C.c_int), scal, C.c_int, C.c_int, C.c_int, dp, C.c_int]
        _handles[key] = h
    return _handles[key]


class JointCpu:
    """Host-side state of one joint fit: data, 1 / sigma^2 (E, n, n), narrow PSFs (E, N, N), parameters, moments."""

    def __init__(self, data, sigma2, psf, ss, M, double=False, threads=0, lib_path=None):
        self.double = bool(double)
        self.dt = np.float64 if double else np.float32
        self.l = lib(double, lib_path)
        self.ptr = C.POINTER(C.c_double if double else C.c_float)
        self.data = np.ascontiguousarray(data, self.dt)
        s2 = np.asarray(sigma2, np.float64)
        self.wgt = np.ascontiguousarray(np.where(np.isfinite(s2) & (s2 > 0), 1.0 / s2, 0.0), self.dt)
        psf = np.ascontiguousarray(psf, self.dt)
        self.E, self.n, _ = self.data.shape
        self.ss, self.M = int(ss), int(M)
        self.N = self.n * self.ss
        self.J = int(np.log2(self.N))
        E, M, NN = self.E, self.M, self.N * self.N
        self.h = self.l.jc_cpu_create(E, M, self.n, self.ss, self._p(self.data), self._p(self.wgt), self._p(psf), int(threads))
        if not self.h:
            raise RuntimeError('jc_cpu_create failed')
        self.p = dict(a=np.zeros(E * M, self.dt), c_x=np.zeros(M, self.dt), c_y=np.zeros(M, self.dt), dx=np.zeros(E, self.dt),
                      dy=np.zeros(E, self.dt), alpha=np.zeros(E, self.dt), mean=np.zeros(E, self.dt), h=np.zeros(NN, self.dt))
        self.mom_m = np.zeros(E * M + 2 * M + 3 * E + NN, self.dt)
        self.mom_s = np.zeros_like(self.mom_m)
        self.t = 0
        self.W = None
        self.loss_cfg = ((C.c_double if self.double else C.c_float) * 6)()

    def _p(self, a):
        return None if a is None else a.ctypes.data_as(self.ptr)

    def close(self):
        if getattr(self, 'h', None):
            self.l.jc_cpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, **kw):
        for k, v in kw.items():
            self.p[k] = np.array(np.asarray(v, np.float64).reshape(self.p[k].shape), dtype=self.dt, order='C', copy=True)

    def set_loss(self, W=None, lam_scales=0.0, lam_hf=0.0, lam_positivity=0.0, lam_positivity_ps=0.0, lam_pts_source=0.0,
                 lam_flux_uniformity=0.0):
        NN = self.N * self.N
        self.W = None if W is None else np.ascontiguousarray(np.asarray(W, np.float64)[:self.J].reshape(self.J, NN), self.dt)
        for i, v in enumerate((lam_scales, lam_hf, lam_positivity, lam_positivity_ps, lam_pts_source, lam_flux_uniformity)):
            self.loss_cfg[i] = v

    def eval(self, threads=0, want_model=False):
        """-> (loss, dict of gradients [, model])"""
        E, M, n, NN = self.E, self.M, self.n, self.N * self.N
        g = dict(a=np.empty(E * M, self.dt), c_x=np.empty(M, self.dt), c_y=np.empty(M, self.dt), dx=np.empty(E, self.dt),
                 dy=np.empty(E, self.dt), mean=np.empty(E, self.dt), h=np.empty(NN, self.dt))
        model = np.empty((E, n, n), self.dt) if want_model else None
        loss = C.c_double()
        p = self.p
        rc = self.l.jc_cpu_eval(self.h, C.cast(self.loss_cfg, C.c_void_p), self._p(self.W), self._p(p['a']), self._p(p['c_x']),
                                self._p(p['c_y']), self._p(p['dx']), self._p(p['dy']), self._p(p['alpha']), self._p(p['mean']),
                                self._p(p['h']), C.byref(loss), self._p(g['a']), self._p(g['c_x']), self._p(g['c_y']),
                                self._p(g['dx']), self._p(g['dy']), self._p(g['mean']), self._p(g['h']), self._p(model), int(threads))
        if rc:
            raise RuntimeError(f'jc_cpu_eval failed ({rc})')
        return (loss.value, g, model) if want_model else (loss.value, g)

    def run(self, n_iter, lr0=1e-4, schedule=False, free=FREE_ORDER, threads=0):
        """n_iter AdaBelief iterations; returns the loss history (n_iter + 1: before every update, then the final loss)."""
        hist = np.empty(n_iter + 1, np.float64)
        p = self.p
        mask = (C.c_int * 7)(*[1 if k in free else 0 for k in FREE_ORDER])
        lr = C.c_double(lr0) if self.double else C.c_float(lr0)
        rc = self.l.jc_cpu_run(self.h, C.cast(self.loss_cfg, C.c_void_p), self._p(self.W), self._p(p['a']), self._p(p['c_x']),
                               self._p(p['c_y']), self._p(p['dx']), self._p(p['dy']), self._p(p['alpha']), self._p(p['mean']),
                               self._p(p['h']), self._p(self.mom_m), self._p(self.mom_s), mask, lr, int(bool(schedule)), self.t,
                               int(n_iter), hist.ctypes.data_as(C.POINTER(C.c_double)), int(threads))
        if rc:
            raise RuntimeError(f'jc_cpu_run failed ({rc})')
        self.t += int(n_iter)
        return hist
