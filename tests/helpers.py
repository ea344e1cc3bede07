"""Shared helpers of the parity tests: oracle-side parameter packing."""
import math

import numpy as np
import torch

from oracle import model as om

# wall-clock ratios measured inside correctness tests; asserted by tests/test_zz_perf_gpu.py only (marker `perf`)
PERF = {}


def psf_oracle_inputs(ds, f, ss):
    data = om.T(ds['data'][f])
    sig2 = om.T(ds['noisemap'][f]) ** 2
    mask = om.T(ds['masks'][f].astype(np.float64))
    return data, sig2, mask


def psf_initial_params(ds, f, ss, rng=None, jitter=0.0):
    """Stage-A style starting point (Moffat from the seeing guess, flux = masked stamp sum)."""
    S, n, _ = ds['data'][f].shape
    N = ss * n
    g = float(ds['fwhm_guess'][f])
    f0 = math.sqrt(max(g * g - (2.0 / ss) ** 2, 1.0))
    a = (ds['data'][f] * ds['masks'][f]).sum(axis=(-1, -2)).astype(np.float64)
    p = dict(fwhm_x=f0, fwhm_y=0.9 * f0, phi=0.3, beta=2.5, B=np.zeros(N * N), a=a,
             x0=np.zeros(S), y0=np.zeros(S), sky=np.zeros(S))
    if rng is not None and jitter > 0:
        p['x0'] = rng.uniform(-jitter, jitter, S)
        p['y0'] = rng.uniform(-jitter, jitter, S)
        p['B'] = 1e-4 * rng.standard_normal(N * N)
        p['sky'] = 1e-3 * rng.standard_normal(S)
    return {k: om.T(v) for k, v in p.items()}


def weights_from(ds):
    """mask / sigma^2 as the C ABI expects it."""
    return (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)


def stars_array(plist):
    """[F][S][4] = a, x0, y0, sky from a list of oracle parameter dicts."""
    return np.stack([np.stack([p['a'].numpy(), p['x0'].numpy(), p['y0'].numpy(), p['sky'].numpy()], axis=-1)
                     for p in plist]).astype(np.float32)


def moffat_array(plist):
    return np.stack([[float(p['fwhm_x']), float(p['fwhm_y']), float(p['phi']), float(p['beta'])]
                     for p in plist]).astype(np.float32)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
