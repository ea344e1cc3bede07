"""ctypes front of oracle/joint_ps_cpu.c (C / OpenMP restatement of the point-source-only joint fit: star photometry).

TEST / MEASUREMENT INFRASTRUCTURE ONLY - see ``oracle/__init__.py``: bench.py's ``cpu_baseline`` leg of the star-photometry
entry times it, tests/ use it as a second checker.  Nothing under ``lightcurver_amd/`` imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {False: os.path.join(_HERE, '_build', 'libjointpscpu.so'), True: os.path.join(_HERE, '_build', 'libjointpscpu_f64.so')}
_handles = {}


def build(double=False, native=False, out=None):
    """gcc -O3 -fopenmp oracle/joint_ps_cpu.c -> oracle/_build/libjointpscpu[_f64].so (x86-64-v3: the file built in the
    build container also runs on the GPU box's host CPU)."""
    out = out or _LIBS[bool(double)]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    arch = 'native' if native else 'x86-64-v3'
    subprocess.run(['gcc', '-O3', f'-march={arch}', '-fopenmp', '-fPIC', '-std=c11', '-shared'] +
                   (['-DJPS_CPU_DOUBLE'] if double else []) + [os.path.join(_HERE, 'joint_ps_cpu.c'), '-o', out, '-lm'], check=True)
    return out


def lib(double=False):
    double = bool(double)
    if double not in _handles:
        if not os.path.exists(_LIBS[double]):
            build(double)
        h = C.CDLL(_LIBS[double])
        ptr = C.POINTER(C.c_double if double else C.c_float)
        scal = C.c_double if double else C.c_float
        dp = C.POINTER(C.c_double)
        h.jps_cpu_eval.restype = C.c_int
        h.jps_cpu_eval.argtypes = [C.c_int] * 4 + [ptr] * 9 + [dp] + [ptr] * 7 + [C.c_int]
        h.jps_cpu_run.restype = C.c_int
        h.jps_cpu_run.argtypes = [C.c_int] * 4 + [ptr] * 11 + [C.c_int, scal, C.c_int, C.c_int, C.c_int, dp, C.c_int]
        _handles[double] = h
    return _handles[double]


class JointPsCpu:
    """Host-side state of one star-photometry fit: data, 1 / sigma^2 (E, n, n), narrow PSFs (E, N, N), parameters, moments."""

    def __init__(self, data, sigma2, psf, ss, M=1, double=False):
        self.double = bool(double)
        self.dt = np.float64 if double else np.float32
        self.l = lib(double)
        self.ptr = C.POINTER(C.c_double if double else C.c_float)
        self.data = np.ascontiguousarray(data, self.dt)
        self.wgt = np.ascontiguousarray(1.0 / np.asarray(sigma2, np.float64), self.dt)
        self.psf = np.ascontiguousarray(psf, self.dt)
        self.E, self.n, _ = self.data.shape
        self.ss, self.M = int(ss), int(M)
        E, M = self.E, self.M
        self.p = dict(a=np.zeros(E * M, self.dt), c_x=np.zeros(M, self.dt), c_y=np.zeros(M, self.dt), dx=np.zeros(E, self.dt),
                      dy=np.zeros(E, self.dt), mean=np.zeros(E, self.dt))
        self.mom_m = np.zeros(E * M + 2 * M + 3 * E, self.dt)
        self.mom_s = np.zeros_like(self.mom_m)
        self.t = 0

    def _p(self, a):
        return a.ctypes.data_as(self.ptr)

    def set_params(self, **kw):
        for k, v in kw.items():
            self.p[k] = np.array(np.asarray(v, np.float64).reshape(self.p[k].shape), dtype=self.dt, order='C', copy=True)   # run() steps it in place

    def eval(self, threads=0, want_model=False):
        """-> (loss, dict of gradients [, model])"""
        E, M, n = self.E, self.M, self.n
        g = dict(a=np.empty(E * M, self.dt), c_x=np.empty(M, self.dt), c_y=np.empty(M, self.dt), dx=np.empty(E, self.dt),
                 dy=np.empty(E, self.dt), mean=np.empty(E, self.dt))
        model = np.empty((E, n, n), self.dt) if want_model else None
        loss = C.c_double()
        p = self.p
        rc = self.l.jps_cpu_eval(E, M, n, self.ss, self._p(self.data), self._p(self.wgt), self._p(self.psf), self._p(p['a']),
                                 self._p(p['c_x']), self._p(p['c_y']), self._p(p['dx']), self._p(p['dy']), self._p(p['mean']),
                                 C.byref(loss), self._p(g['a']), self._p(g['c_x']), self._p(g['c_y']), self._p(g['dx']),
                                 self._p(g['dy']), self._p(g['mean']), self._p(model) if want_model else None, int(threads))
        if rc:
            raise RuntimeError(f'jps_cpu_eval failed ({rc})')
        return (loss.value, g, model) if want_model else (loss.value, g)

    def run(self, n_iter, lr0=1e-3, schedule=True, free_mean=False, threads=0):
        """n_iter AdaBelief iterations; returns the loss history (n_iter + 1: before every update, then the final loss)."""
        hist = np.empty(n_iter + 1, np.float64)
        p = self.p
        lr = C.c_double(lr0) if self.double else C.c_float(lr0)
        rc = self.l.jps_cpu_run(self.E, self.M, self.n, self.ss, self._p(self.data), self._p(self.wgt), self._p(self.psf),
                                self._p(p['a']), self._p(p['c_x']), self._p(p['c_y']), self._p(p['dx']), self._p(p['dy']),
                                self._p(p['mean']), self._p(self.mom_m), self._p(self.mom_s), int(bool(free_mean)), lr,
                                int(bool(schedule)), self.t, int(n_iter), hist.ctypes.data_as(C.POINTER(C.c_double)), int(threads))
        if rc:
            raise RuntimeError(f'jps_cpu_run failed ({rc})')
        self.t += int(n_iter)
        return hist
