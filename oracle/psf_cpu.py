"""ctypes front of oracle/psf_cpu.c (fp32 C / OpenMP restatement of the PSF pixel-grid stage).

TEST / MEASUREMENT INFRASTRUCTURE ONLY - see ``oracle/__init__.py``: bench.py's ``cpu_baseline`` leg times it,
tests/ use it as a second checker.  Nothing under ``lightcurver_amd/`` imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, '_build', 'libpsfcpu.so')
_LIB64 = os.path.join(_HERE, '_build', 'libpsfcpu_f64.so')
_handle = None
_handle64 = None
fp = C.POINTER(C.c_float)
dp = C.POINTER(C.c_double)


def build(native=False, out=None, double=False):
    """gcc -O3 -fopenmp oracle/psf_cpu.c -> oracle/_build/libpsfcpu.so (x86-64-v3 so that the file built in the
    build container also runs on the GPU box's host CPU); native=True compiles for the CPU it runs on; double=True
    builds the real = double variant."""
    out = out or (_LIB64 if double else _LIB)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    arch = 'native' if native else 'x86-64-v3'
    subprocess.run(['gcc', '-O3', f'-march={arch}', '-fopenmp', '-fPIC', '-std=c11', '-shared'] +
                   (['-DPSF_CPU_DOUBLE'] if double else []) +
                   [os.path.join(_HERE, 'psf_cpu.c'), '-o', out, '-lm'], check=True)
    return out


def lib(path=None, double=False):
    global _handle, _handle64
    if path is not None:
        h = C.CDLL(path)
    elif double:
        if _handle64 is None:
            if not os.path.exists(_LIB64):
                build(double=True)
            _handle64 = C.CDLL(_LIB64)
        h = _handle64
    else:
        if _handle is None:
            if not os.path.exists(_LIB):
                build()
            _handle = C.CDLL(_LIB)
        h = _handle
    ptr, scal = (dp, C.c_double) if double else (fp, C.c_float)
    h.psf_cpu_run.restype = C.c_int
    h.psf_cpu_run.argtypes = [C.c_int] * 4 + [ptr] * 10 + [scal] * 3 + [C.c_int] * 3 + [ptr, C.c_int]
    h.psf_cpu_eval.restype = C.c_int
    h.psf_cpu_eval.argtypes = [C.c_int] * 4 + [ptr] * 6 + [scal] * 2 + [ptr] * 5
    return h


class PsfCpuState:
    """Host-side state of one batch: data, weight [F][S][n][n], Moffat Tm [F][N*N], W [F][J][N*N], B and stars."""

    def __init__(self, data, weight, ss, Tm, W, B, stars, handle=None, double=False):
        self.dtype = np.float64 if double else np.float32
        ptr = dp if double else fp
        _f = lambda a: np.ascontiguousarray(a, dtype=self.dtype)
        self._p = lambda a: None if a is None else a.ctypes.data_as(ptr)
        self.data, self.wgt = _f(data), _f(weight)
        self.F, self.S, self.n, _ = self.data.shape
        self.ss = int(ss)
        self.N = self.n * self.ss
        self.J = int(np.log2(self.N))
        NN = self.N * self.N
        self.Tm = _f(Tm).reshape(self.F, NN)
        self.W = _f(W).reshape(self.F, -1, NN)[:, :self.J].copy()
        self.B = _f(B).reshape(self.F, NN).copy()
        self.mB, self.sB = np.zeros_like(self.B), np.zeros_like(self.B)
        self.stars = _f(stars).reshape(self.F, self.S, 4).copy()
        self.stars_m, self.stars_s = np.zeros_like(self.stars), np.zeros_like(self.stars)
        self.t = 0
        self._l = handle or lib(double=double)

    def evaluate(self, lam_sc=1.0, lam_hf=1.0, model=False):
        F, S, n, N = self.F, self.S, self.n, self.N
        _p, dt = self._p, self.dtype
        out = dict(loss=np.empty(F, dt), chi2=np.empty(F, dt),
                   grad_grid=np.empty((F, N, N), dt), grad_stars=np.empty((F, S, 3), dt))
        mod = np.empty((F, S, n, n), dt) if model else None
        rc = self._l.psf_cpu_eval(F, S, n, self.ss, _p(self.data), _p(self.wgt), _p(self.Tm), _p(self.W), _p(self.B),
                                  _p(self.stars), lam_sc, lam_hf, _p(out['loss']), _p(out['chi2']), _p(out['grad_grid']),
                                  _p(out['grad_stars']), _p(mod))
        if rc:
            raise MemoryError('psf_cpu_eval')
        if model:
            out['model'] = mod
        return out

    def run_adabelief(self, n_iter, lr0=1e-4, schedule=True, lam_sc=1.0, lam_hf=1.0, threads=0):
        """-> loss history (F, n_iter + 1): loss before every update, then the final loss."""
        hist = np.empty((self.F, n_iter + 1), self.dtype)
        _p = self._p
        rc = self._l.psf_cpu_run(self.F, self.S, self.n, self.ss, _p(self.data), _p(self.wgt), _p(self.Tm), _p(self.W),
                                 _p(self.B), _p(self.mB), _p(self.sB), _p(self.stars), _p(self.stars_m), _p(self.stars_s),
                                 lam_sc, lam_hf, lr0, 1 if schedule else 0, self.t, int(n_iter), _p(hist), int(threads))
        if rc:
            raise MemoryError('psf_cpu_run')
        self.t += int(n_iter)
        return hist
