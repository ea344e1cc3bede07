"""``build_psf`` on the MI355X: the call lightcurver makes once per frame
(reference lightcurver/processes/psf_modelling.py:164-171; structural contract in
tests/test_starred_calls/test_starred_calls.py:66-80), plus ``build_psf_batch`` which fits every
frame of a dataset in one device batch (one workgroup per frame) instead of the reference's serial
Python loop (psf_modelling.py:92)."""
import math

import numpy as np

from ... import _lib
from ...psf_batch import PsfBatch


def _initial_positions(data, masks, method):
    """Initial (x0, y0) in data pixels relative to the stamp centre."""
    S, n, _ = data.shape
    if method == 'center':
        return np.zeros(S), np.zeros(S)
    c = (n - 1) / 2.0
    yy, xx = np.mgrid[0:n, 0:n]
    x0 = np.zeros(S)
    y0 = np.zeros(S)
    for i in range(S):
        img = np.where(masks[i] > 0, data[i], 0.0)
        if method == 'max':
            iy, ix = np.unravel_index(np.argmax(img), img.shape)
            x0[i], y0[i] = ix - c, iy - c
        elif method == 'barycenter':
            pos = np.clip(img, 0.0, None)
            tot = pos.sum()
            if tot > 0:
                x0[i], y0[i] = (pos * xx).sum() / tot - c, (pos * yy).sum() / tot - c
        else:
            raise ValueError(f'unknown guess_method_star_position {method!r}')
    lim = n / 4.0
    return np.clip(x0, -lim, lim), np.clip(y0, -lim, lim)


def build_psf_batch(images, noisemaps, subsampling_factor, masks=None, n_iter_analytic=40,
                    n_iter_adabelief=2000, guess_method_star_position='barycenter', guess_fwhm_pixels=3.,
                    field_distortion=False, stamp_coordinates=None, regularization_strength_scales=1.,
                    regularization_strength_hf=1., init_learning_rate=1e-4, schedule_learning_rate=True,
                    device=0, ctx=None):
    """Fit the PSF of every frame of a list: images[f] is (S_f, n, n) (ragged S_f allowed, the
    reference drops heavily masked stamps per frame at psf_modelling.py:144-153).

    Returns one result dict per frame with the keys of STARRED's ``build_psf``.
    """
    if field_distortion:
        raise NotImplementedError('field_distortion=True is not built yet (DESIGN.md, out-of-scope list)')
    F = len(images)
    if F == 0:
        return []
    ss = int(subsampling_factor)
    n = int(np.asarray(images[0]).shape[-1])
    S_list = [int(np.asarray(im).shape[0]) for im in images]
    if min(S_list) < 1:
        raise ValueError('every frame needs at least one stamp')
    S = max(S_list)
    N = n * ss
    data = np.zeros((F, S, n, n), np.float64)
    weight = np.zeros((F, S, n, n), np.float64)
    norms = np.ones(F)
    stars = np.zeros((F, S, 4), np.float64)
    fw = np.broadcast_to(np.asarray(guess_fwhm_pixels, dtype=np.float64), (F,))
    moffat = np.zeros((F, 4))
    for f in range(F):
        img = np.array(images[f], dtype=np.float64)
        noi = np.array(noisemaps[f], dtype=np.float64)
        if img.shape != noi.shape or img.shape[1:] != (n, n):
            raise ValueError('image / noisemap shape mismatch')
        msk = np.ones_like(img) if masks is None or masks[f] is None else np.asarray(masks[f], dtype=np.float64)
        bad = ~(np.isfinite(img) & np.isfinite(noi)) | (noi <= 0)
        msk = np.where(bad, 0.0, msk)
        img = np.where(bad, 0.0, img)
        noi = np.where(bad, 1.0, noi)
        norm = float(np.max(img * (msk > 0))) if np.any(msk > 0) else 1.0
        if not np.isfinite(norm) or norm <= 0:
            norm = 1.0
        norms[f] = norm
        Sf = S_list[f]
        data[f, :Sf] = img / norm
        weight[f, :Sf] = (msk > 0) / (noi / norm) ** 2
        x0, y0 = _initial_positions(data[f, :Sf], msk, guess_method_star_position)
        stars[f, :Sf, 0] = np.clip((data[f, :Sf] * (msk > 0)).sum(axis=(-1, -2)), 1e-6, None)
        stars[f, :Sf, 1] = x0
        stars[f, :Sf, 2] = y0
        f0 = math.sqrt(max(float(fw[f]) ** 2 - (2.0 / ss) ** 2, (1.0 / ss) ** 2))
        moffat[f] = (f0, f0, 0.0, 2.5)

    ctx = ctx or _lib.default_context(device)
    b = PsfBatch(data, weight, ss, ctx)
    try:
        b.set_moffat(moffat)
        b.set_stars(stars)
        b.set_grid(None)
        # stage A: elliptical Moffat + amplitudes + positions, pixel grid fixed to zero (L-BFGS)
        analytic_loss = b.fit_moffat(int(n_iter_analytic)) if n_iter_analytic > 0 else None
        # stage B: free the pixel grid, l1-starlet with noise-propagated weights (AdaBelief)
        b.propagate_noise()
        b.set_regularization(None, float(regularization_strength_scales), float(regularization_strength_hf))
        if n_iter_adabelief > 0:
            b.run_adabelief(int(n_iter_adabelief), init_learning_rate=init_learning_rate,
                            schedule_learning_rate=schedule_learning_rate)
        hist = b.loss_history()
        res = b.results()
        mof = b.get_moffat().astype(np.float64)
        st = b.get_stars().astype(np.float64)
        grid = b.get_grid()
    finally:
        b.close()

    out = []
    for f in range(F):
        Sf = S_list[f]
        resid = res['residuals'][f, :Sf].astype(np.float64) * norms[f]
        kwargs_psf = {
            'kwargs_moffat': {'fwhm_x': np.array([mof[f, 0]]), 'fwhm_y': np.array([mof[f, 1]]),
                              'phi': np.array([mof[f, 2]]), 'beta': np.array([mof[f, 3]]), 'C': np.array([1.0])},
            'kwargs_gaussian': {'a': st[f, :Sf, 0] * norms[f], 'x0': st[f, :Sf, 1], 'y0': st[f, :Sf, 2]},
            'kwargs_background': {'background': grid[f].reshape(N * N), 'mean': st[f, :Sf, 3] * norms[f]},
            'kwargs_distortion': {},
        }
        out.append({
            'full_psf': res['full_psf'][f],
            'narrow_psf': res['narrow_psf'][f],
            'models': np.asarray(images[f], dtype=np.float64) - resid,
            'residuals': resid,
            'kwargs_psf': kwargs_psf,
            'chi2': float(res['chi2'][f]),
            'norm': norms[f],
            'analytic_extra_fields': {'final_loss': None if analytic_loss is None else float(analytic_loss[f])},
            'adabelief_extra_fields': {'loss_history': [float(v) for v in hist[f, 1:]],
                                       'initial_loss': float(hist[f, 0])},
        })
    return out


def build_psf(image, noisemap, subsampling_factor, masks=None, n_iter_analytic=40, n_iter_adabelief=2000,
              guess_method_star_position='barycenter', guess_fwhm_pixels=3., field_distortion=False,
              stamp_coordinates=None, **kwargs):
    """Drop-in for ``starred.procedures.psf_routines.build_psf`` (one frame, S stamps)."""
    return build_psf_batch([image], [noisemap], subsampling_factor, masks=[masks],
                           n_iter_analytic=n_iter_analytic, n_iter_adabelief=n_iter_adabelief,
                           guess_method_star_position=guess_method_star_position,
                           guess_fwhm_pixels=guess_fwhm_pixels, field_distortion=field_distortion,
                           stamp_coordinates=stamp_coordinates, **kwargs)[0]
