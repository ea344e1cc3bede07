"""Stamp pre-processing on the device: one fused pass over a stack of stamps that
replaces the NumPy clean-up the reference runs on the host before its fits
(lightcurver/processes/cutout_making.py:43-51, psf_modelling.py:135-153, roi_file_preparation.py:162-201,
star_photometry.py:309-316).  Thin wrapper over ``lc_prepare_stamps`` (include/lcmi.h)."""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import f32, ptr


def prepare_stamps(data, noisemap=None, rms=None, exptime=None, coefficient=None, bad=None, nan_noise=1.0,
                   noise_boost=0.0, boost_whole_stamp=False, ctx=None, want=('data', 'noisemap', 'weight')):
    """data (K, n, n) float; noisemap (K, n, n) or None (then rms (K,), exptime (K,) build it);
    coefficient (K,) or None; bad (K, n, n) bool, True = flagged, or None.

    Returns dict with the requested arrays among data / noisemap / weight (float32, same shape as data),
    'masked_count' (K,) int32 and 'kernel_ms' (device time of the kernel)."""
    ctx = ctx or _lib.default_context()
    lib = _lib.lib()
    d = f32(data)
    K = d.shape[0]
    npix = int(np.prod(d.shape[1:]))
    nm = f32(noisemap) if noisemap is not None else None
    r = f32(np.broadcast_to(rms, (K,))) if rms is not None else None
    t = f32(np.broadcast_to(exptime, (K,))) if exptime is not None else None
    c = f32(np.broadcast_to(coefficient, (K,))) if coefficient is not None else None
    b = np.ascontiguousarray(np.asarray(bad).astype(np.uint8)) if bad is not None else None
    if nm is not None and nm.shape != d.shape or b is not None and b.shape != d.shape:
        raise ValueError('noisemap / bad must have the shape of data')
    out = {k: np.empty(d.shape, np.float32) for k in want}
    count = np.empty(K, np.int32)
    ms = C.c_float()
    bp = b.ctypes.data_as(C.POINTER(C.c_uint8)) if b is not None else None
    ctx.check(lib.lc_prepare_stamps(ctx.h, K, npix, ptr(d), ptr(nm), ptr(r), ptr(t), ptr(c), bp, float(nan_noise),
                                    float(noise_boost), int(bool(boost_whole_stamp)), ptr(out.get('data')),
                                    ptr(out.get('noisemap')), ptr(out.get('weight')),
                                    count.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ms)), 'lc_prepare_stamps')
    out['masked_count'] = count
    out['kernel_ms'] = ms.value
    return out
