"""Seeded synthetic stamp stacks for benchmarks, smoke runs and parity tests.

There is no network and no survey data on the build or GPU boxes, so every workload of
BASELINE.json (C1..C5) is generated on the fly from ``numpy.random.default_rng(seed)`` following
SURVEY.md section 8(d): elliptical Moffat PSFs (beta = 3, FWHM 3-5 data px) with a 2 % smooth
perturbation, star fluxes log-uniform in [1e3, 1e5] e-, noise sigma^2 = rms^2 + |data| with
rms = 5 e- (the noise model of the reference's cutout_making.py:43-51), 1 % masked pixels, values
rescaled to O(1).  This is data generation (NumPy/SciPy on the host), not part of the hot path.
"""
import math

import numpy as np
from scipy.ndimage import gaussian_filter, map_coordinates
from scipy.signal import fftconvolve

GAUSS_FWHM = 2.0
SIGMA_G = GAUSS_FWHM / (2.0 * math.sqrt(2.0 * math.log(2.0)))

CONFIGS = {
    'C1': dict(kind='psf', F=10, S=4, n=32, ss=2, seed=101),
    'C2': dict(kind='psf', F=100, S=8, n=32, ss=2, seed=102),
    'C3': dict(kind='psf', F=500, S=8, n=64, ss=2, seed=103),
    'C4': dict(kind='roi', E=200, M=2, n=64, ss=2, seed=104),
    'C5': dict(kind='roi', E=1000, M=4, n=128, ss=2, seed=105),
}


def _gauss2d(N, X, Y):
    v = np.arange(N, dtype=np.float64)
    gx = np.exp(-0.5 * ((v - X) / SIGMA_G) ** 2)
    gy = np.exp(-0.5 * ((v - Y) / SIGMA_G) ** 2)
    return np.outer(gy, gx) / (2.0 * math.pi * SIGMA_G ** 2)


def _moffat(N, ss, fwhm_x, fwhm_y, phi, beta):
    c = (N - 1) // 2
    idx = np.arange(N, dtype=np.float64) - c
    x, y = idx[None, :], idx[:, None]
    k = 2.0 * math.sqrt(2.0 ** (1.0 / beta) - 1.0)
    ax, ay = ss * fwhm_x / k, ss * fwhm_y / k
    xr = x * math.cos(phi) + y * math.sin(phi)
    yr = -x * math.sin(phi) + y * math.cos(phi)
    m = (1.0 + (xr / ax) ** 2 + (yr / ay) ** 2) ** (-beta)
    return m / m.sum()


def _blocksum(img, ss):
    n = img.shape[-1] // ss
    return img.reshape(img.shape[:-2] + (n, ss, n, ss)).sum(axis=(-1, -3))


def make_narrow_psf(rng, N, ss, perturb=0.02):
    """One 'true' narrow PSF (unit sum) and the parameters it was drawn from."""
    fwhm_full = rng.uniform(3.0, 5.0)
    q = rng.uniform(0.9, 1.0)
    phi = rng.uniform(0.0, math.pi)
    fwhm = math.sqrt(max(fwhm_full ** 2 - (GAUSS_FWHM / ss) ** 2, 1.0))
    mof = _moffat(N, ss, fwhm, fwhm * q, phi, 3.0)
    pert = gaussian_filter(rng.standard_normal((N, N)), 2.0)
    pert *= perturb * mof.max() / np.abs(pert).max()
    # keep the perturbation where the PSF has signal so the truth stays positive and compact
    pert *= mof / mof.max() * 4.0
    s = mof + pert
    return s / s.sum(), dict(fwhm_x=fwhm, fwhm_y=fwhm * q, phi=phi, beta=3.0, fwhm_full=fwhm_full)


def make_psf_dataset(F, S, n, ss=2, seed=0, rms=5.0, masked_fraction=0.01, dtype=np.float32):
    """F frames of S star stamps (n x n): returns dict with data, noisemap, masks (F,S,n,n),
    fwhm_guess (F,), and the truth (narrow PSFs, fluxes, offsets)."""
    rng = np.random.default_rng(seed)
    N = ss * n
    c0 = (N - 1) / 2.0
    data = np.zeros((F, S, n, n))
    noise = np.zeros((F, S, n, n))
    narrow = np.zeros((F, N, N))
    flux = np.exp(rng.uniform(math.log(1e3), math.log(1e5), size=(F, S)))
    x0 = rng.uniform(-0.5, 0.5, size=(F, S))
    y0 = rng.uniform(-0.5, 0.5, size=(F, S))
    fwhm_guess = np.zeros(F)
    for f in range(F):
        narrow[f], par = make_narrow_psf(rng, N, ss)
        fwhm_guess[f] = par['fwhm_full'] * rng.uniform(0.9, 1.1)
        for s in range(S):
            g = _gauss2d(N, c0 + ss * x0[f, s], c0 + ss * y0[f, s])
            clean = flux[f, s] * _blocksum(fftconvolve(g, narrow[f], mode='same'), ss)
            sig = np.sqrt(rms ** 2 + np.abs(clean))
            data[f, s] = clean + sig * rng.standard_normal((n, n))
            noise[f, s] = np.sqrt(rms ** 2 + np.abs(data[f, s]))
    masks = rng.uniform(size=(F, S, n, n)) >= masked_fraction
    scale = np.percentile(data, 99.9)
    return dict(data=(data / scale).astype(dtype), noisemap=(noise / scale).astype(dtype), masks=masks,
                fwhm_guess=fwhm_guess, scale=scale, ss=ss,
                truth=dict(narrow_psf=narrow, flux=flux / scale, x0=x0, y0=y0))


def make_roi_dataset(E, M, n, ss=2, seed=0, rms=5.0, with_background=True, shift_sigma=0.3,
                     alpha_sigma=0.0, dtype=np.float32):
    """E epochs of an n x n ROI with M point sources, a smooth background and one narrow PSF per
    epoch.  Returns data, noisemap (E,n,n), psf (E,N,N), truth parameters (STARRED kwargs layout:
    a epoch-major, c_x, c_y, dx, dy, alpha, h flat, mean)."""
    rng = np.random.default_rng(seed)
    N = ss * n
    c0 = (N - 1) / 2.0
    c_x = rng.uniform(-n / 4.0, n / 4.0, size=M)
    c_y = rng.uniform(-n / 4.0, n / 4.0, size=M)
    base = np.exp(rng.uniform(math.log(2e3), math.log(5e4), size=M))
    phase = rng.uniform(0, 2 * math.pi, size=M)
    e = np.arange(E)
    a = base[None, :] * (1.0 + 0.1 * np.sin(2 * math.pi * e[:, None] / max(E, 1) + phase[None, :]))
    dx = rng.normal(0.0, shift_sigma, size=E)
    dy = rng.normal(0.0, shift_sigma, size=E)
    dx[0] = dy[0] = 0.0
    alpha = rng.normal(0.0, alpha_sigma, size=E) if alpha_sigma > 0 else np.zeros(E)
    alpha[0] = 0.0
    h = np.zeros((N, N))
    if with_background:
        u = np.arange(N, dtype=np.float64)
        for _ in range(3):
            bx, by = rng.uniform(0.3 * N, 0.7 * N, size=2)
            bs = rng.uniform(0.08 * N, 0.2 * N)
            amp = rng.uniform(5.0, 20.0)
            h += amp * np.exp(-0.5 * (((u[None, :] - bx) / bs) ** 2 + ((u[:, None] - by) / bs) ** 2))
    psf = np.zeros((E, N, N))
    data = np.zeros((E, n, n))
    noise = np.zeros((E, n, n))
    idx = np.arange(N, dtype=np.float64)
    for k in range(E):
        psf[k], _ = make_narrow_psf(rng, N, ss)
        ca, sa = math.cos(math.radians(alpha[k])), math.sin(math.radians(alpha[k]))
        scene = np.zeros((N, N))
        for i in range(M):
            X = c0 + ss * (ca * c_x[i] - sa * c_y[i] + dx[k])
            Y = c0 + ss * (sa * c_x[i] + ca * c_y[i] + dy[k])
            scene += a[k, i] * _gauss2d(N, X, Y)
        px = (idx - c0)[None, :] - ss * dx[k]
        py = (idx - c0)[:, None] - ss * dy[k]
        Xs = c0 + ca * px + sa * py
        Ys = c0 - sa * px + ca * py
        scene += map_coordinates(h, [Ys, Xs], order=1, mode='nearest')
        clean = _blocksum(fftconvolve(scene, psf[k], mode='same'), ss)
        sig = np.sqrt(rms ** 2 + np.abs(clean))
        data[k] = clean + sig * rng.standard_normal((n, n))
        noise[k] = np.sqrt(rms ** 2 + np.abs(data[k]))
    scale = np.max(data)
    truth = dict(a=(a / scale).reshape(E * M), c_x=c_x, c_y=c_y, dx=dx, dy=dy, alpha=alpha,
                 h=(h / scale).reshape(N * N), mean=np.zeros(E))
    return dict(data=(data / scale).astype(dtype), noisemap=(noise / scale).astype(dtype),
                psf=psf.astype(dtype), ss=ss, scale=scale, truth=truth)


def make_config(name, **overrides):
    cfg = dict(CONFIGS[name])
    cfg.update(overrides)
    kind = cfg.pop('kind')
    if kind == 'psf':
        return make_psf_dataset(**cfg)
    return make_roi_dataset(**cfg)
