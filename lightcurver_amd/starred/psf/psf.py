"""``apply_distortion`` (reference call sites: lightcurver/processes/star_photometry.py:291-304,
roi_file_preparation.py:169-180): the narrow PSF of a frame resampled at the position of a star / ROI with the field
distortion fitted by ``build_psf(field_distortion=True)``.

Distortion model (unverified against STARRED like the rest of the SPEC):
``kwargs_distortion = {'dilation_x': (3,), 'dilation_y': (3,), 'shear': (3,)}``, each a first-order polynomial
``c0 + c1 x + c2 y`` of the rescaled frame coordinates (x, y) in [-0.5, 0.5]
(lightcurver/utilities/image_coordinates.py:6-27).  The resampling runs on the device (csrc/distort.hip)."""
import numpy as np

from ... import _lib
from ..._lib import f32, ptr

DISTORTION_KEYS = ('dilation_x', 'dilation_y', 'shear')


def distortion_coefficients(kwargs_distortion):
    """The 9 coefficients in the order of the C ABI; missing keys (or an empty dict) mean no distortion."""
    unknown = set(kwargs_distortion or {}) - set(DISTORTION_KEYS)
    if unknown:
        raise KeyError(f'unknown distortion keys {sorted(unknown)}; known: {DISTORTION_KEYS}')
    out = np.zeros((3, 3), np.float64)
    for i, k in enumerate(DISTORTION_KEYS):
        if kwargs_distortion and k in kwargs_distortion:
            v = np.ravel(np.asarray(kwargs_distortion[k], dtype=np.float64))
            if v.size != 3:
                raise ValueError(f'kwargs_distortion[{k!r}] must hold 3 coefficients (c0, c1, c2), got {v.size}')
            out[i] = v
    return out.reshape(9)


def apply_distortion(narrow_psf, kwargs_distortion, star_xy_coordinates, ctx=None):
    """-> (N, N) for one position ((2,) or (1, 2) coordinates), (K, N, N) for K positions."""
    psf = f32(narrow_psf)
    if psf.ndim != 2 or psf.shape[0] != psf.shape[1]:
        raise ValueError('narrow_psf must be (N, N)')
    xy = np.asarray(star_xy_coordinates, dtype=np.float64)
    single = xy.size == 2
    xy = f32(xy.reshape(-1, 2))
    coef = f32(distortion_coefficients(kwargs_distortion))
    ctx = ctx or _lib.default_context()
    N, K = psf.shape[0], xy.shape[0]
    out = np.empty((K, N, N), np.float32)
    ctx.check(_lib.lib().lc_apply_distortion(ctx.h, N, K, ptr(psf), ptr(coef), ptr(xy), ptr(out)), 'lc_apply_distortion')
    return out[0] if single else out
