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


def _fit_size(n, ss):
    """The stamp size the device fits an n x n stamp at: n itself when a kernel is instantiated for it, otherwise the next
    instantiated size of the same parity (the stamp is embedded centrally, psf_routines.build_psf_batch)."""
    l = _lib.lib()
    if l.lc_psf_supported(int(n), int(ss)):
        return int(n)
    for m in range(int(n) + 2, 129, 2):
        if l.lc_psf_supported(m, int(ss)):
            return m
    raise _lib.LcError(f'no PSF kernel for {n}x{n} stamps at subsampling {ss} (instantiated: 16 at ss = 1; 16, 24, 32, 64 at '
                       f'ss = 2; smaller even sizes are fitted embedded in the next of these)')


def build_psf_batch(images, noisemaps, subsampling_factor, masks=None, n_iter_analytic=40,
                    n_iter_adabelief=2000, guess_method_star_position='barycenter', guess_fwhm_pixels=3.,
                    field_distortion=False, stamp_coordinates=None, regularization_strength_scales=1.,
                    regularization_strength_hf=1., init_learning_rate=1e-4, schedule_learning_rate=True,
                    device=0, ctx=None):
    """Fit the PSF of every frame of a list: images[f] is (S_f, n, n) (ragged S_f allowed, the
    reference drops heavily masked stamps per frame at psf_modelling.py:144-153).

    Returns one result dict per frame with the keys of STARRED's ``build_psf``.
    """
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
    # Stamp sizes without a kernel of their own (config.yaml:205 ``stamp_size_stars`` is a free integer) are fitted
    # EMBEDDED in the next instantiated size: the stamps sit in the centre of a larger frame whose extra ring carries zero
    # weight ("absent" to the fit), the pixel grid is fitted on the larger frame and every output is cut back to the
    # caller's size (PSFs renormalised to unit sum).  Same centre, so positions and widths mean the same.
    n_fit = _fit_size(n, ss)
    pad = (n_fit - n) // 2
    if pad:
        big = np.zeros((F, S, n_fit, n_fit), np.float64)
        big[..., pad:pad + n, pad:pad + n] = data
        data = big
        big = np.zeros((F, S, n_fit, n_fit), np.float64)
        big[..., pad:pad + n, pad:pad + n] = weight
        weight = big
    if field_distortion:
        if stamp_coordinates is None:
            raise ValueError('field_distortion=True needs stamp_coordinates (rescaled frame positions of the stamps)')
        coords = stamp_coordinates if F > 1 or np.asarray(stamp_coordinates[0]).ndim == 2 else [stamp_coordinates]
        # (embedded sizes: the two batches run at the fitted size - same centre, zero-weight ring - and the outputs are cut back)
        return _fit_with_distortion(images, data, weight, norms, stars, moffat, S_list, coords, ss, n_fit, ctx,
                                    int(n_iter_analytic), int(n_iter_adabelief), float(regularization_strength_scales),
                                    float(regularization_strength_hf), init_learning_rate, schedule_learning_rate,
                                    n_user=n)
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
    if pad:
        P = pad * ss
        res = dict(res)
        res['residuals'] = res['residuals'][..., pad:pad + n, pad:pad + n]
        for key in ('narrow_psf', 'full_psf'):
            cut = res[key][:, P:P + N, P:P + N]
            res[key] = cut / cut.sum(axis=(-1, -2), keepdims=True)
        grid = np.ascontiguousarray(grid[:, P:P + N, P:P + N])

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
            'adabelief_extra_fields': {'loss_history': hist[f, 1:].astype(np.float64).tolist(),
                                       'initial_loss': float(hist[f, 0])},
        })
    return out


DISTORTION_BOUND = 0.2   # |coefficient| <= 0.2: the resampling kernels assume a distortion close to the identity


def quadratic_forms(theta, xy, ss):
    """(fwhm_x, fwhm_y, phi, beta, 9 distortion coefficients) [..., F, 13] and star coordinates [F][S][2] ->
    q [..., F, S, 4] = (q11, q12, q22, beta) of the Moffat each star sees: Q_i = A_i^-T Q A_i^-1 (csrc/psf_distort.h).
    Leading axes of theta (a batch of parameter sets, e.g. the 26 perturbed copies of a central-difference Jacobian) broadcast."""
    fx, fy, phi, beta = theta[..., 0, None], theta[..., 1, None], theta[..., 2, None], theta[..., 3, None]
    c = theta[..., 4:13]
    x, y = xy[..., 0], xy[..., 1]
    a00 = 1.0 + c[..., 0, None] + c[..., 1, None] * x + c[..., 2, None] * y
    a11 = 1.0 + c[..., 3, None] + c[..., 4, None] * x + c[..., 5, None] * y
    a01 = c[..., 6, None] + c[..., 7, None] * x + c[..., 8, None] * y
    det = a00 * a11 - a01 * a01
    i00, i01, i11 = a11 / det, -a01 / det, a00 / det
    kb = 2.0 * np.sqrt(2.0 ** (1.0 / beta) - 1.0)
    ax, ay = ss * fx / kb, ss * fy / kb
    cs, sn = np.cos(phi), np.sin(phi)
    # Q = R^T diag(1/ax^2, 1/ay^2) R with xr = x cos + y sin, yr = -x sin + y cos
    dx, dy = 1.0 / ax ** 2, 1.0 / ay ** 2
    q11, q12, q22 = dx * cs * cs + dy * sn * sn, (dx - dy) * cs * sn, dx * sn * sn + dy * cs * cs
    # A^-T Q A^-1 (A^-1 symmetric)
    m11 = i00 * q11 + i01 * q12
    m12 = i00 * q12 + i01 * q22
    m21 = i01 * q11 + i11 * q12
    m22 = i01 * q12 + i11 * q22
    out = np.stack([m11 * i00 + m12 * i01, m11 * i01 + m12 * i11, m21 * i01 + m22 * i11, np.broadcast_to(beta, m11.shape)], axis=-1)
    return out


def _fit_with_distortion(images, data, weight, norms, stars, moffat, S_list, coords, ss, n, ctx, n_iter_analytic,
                         n_iter_adabelief, lam_scales, lam_hf, init_learning_rate, schedule_learning_rate, n_user=None):
    """build_psf(field_distortion=True): see include/lcmi.h ("build_psf(field_distortion=True)") for the two-batch scheme.
    n: the size the device fits at; n_user (<= n, same parity): the caller's stamp size when the stamps are embedded."""
    import ctypes as C
    F, S = data.shape[0], data.shape[1]
    N = n * ss
    n_user = n if n_user is None else int(n_user)
    pad = (n - n_user) // 2
    xy = np.zeros((F, S, 2))
    for f in range(F):
        cf = np.asarray(coords[f], dtype=np.float64).reshape(-1, 2)
        if cf.shape[0] != S_list[f]:
            raise ValueError('stamp_coordinates must hold one (x, y) per stamp of the frame')
        xy[f, :S_list[f]] = cf
    star_b = PsfBatch(data.reshape(F * S, 1, n, n), weight.reshape(F * S, 1, n, n), ss, ctx)
    frame_b = PsfBatch(np.zeros((F, 1, n, n)), np.zeros((F, 1, n, n)), ss, ctx)
    lib = _lib.lib()
    try:
        star_b.set_grid(None)
        # ---- stage A: Moffat + distortion + amplitudes / positions, pixel grid zero (batched bounded L-BFGS) ------
        D = 13 + 3 * S
        x = np.zeros((F, D))
        x[:, 0:4] = moffat
        x[:, 13:13 + S] = stars[..., 0]
        x[:, 13 + S:13 + 2 * S] = stars[..., 1]
        x[:, 13 + 2 * S:] = stars[..., 2]
        lo, hi = np.empty((F, D)), np.empty((F, D))
        lo[:, 0:2], hi[:, 0:2] = 0.5 / ss, n_user / 2.0
        lo[:, 2], hi[:, 2] = -math.pi, math.pi
        lo[:, 3], hi[:, 3] = 1.1, 50.0
        lo[:, 4:13], hi[:, 4:13] = -DISTORTION_BOUND, DISTORTION_BOUND
        lo[:, 13:13 + S], hi[:, 13:13 + S] = 0.0, np.inf
        lo[:, 13 + S:], hi[:, 13 + S:] = -n_user / 4.0, n_user / 4.0
        sky = stars[..., 3].copy()

        def push(X):
            q = quadratic_forms(X[:, :13], xy, ss)
            star_b.set_moffat_q(q.reshape(F * S, 4))
            st = np.stack([X[:, 13:13 + S], X[:, 13 + S:13 + 2 * S], X[:, 13 + 2 * S:], sky], axis=-1)
            star_b.set_stars(st.reshape(F * S, 1, 4))
            return q

        failure = []

        def evaluate(_user, Xp, Fp, Gp):
            # (ctypes swallows an exception raised in a callback and returns 0: caught here, reported as a non-zero status so
            #  that lc_batched_lbfgs stops instead of continuing on unwritten values, and re-raised after the call)
            try:
                return _evaluate(Xp, Fp, Gp)
            except BaseException as exc:  # noqa: BLE001
                failure.append(exc)
                return 1

        def _evaluate(Xp, Fp, Gp):
            X = np.ctypeslib.as_array(Xp, shape=(F, D)).copy()
            q = push(X)
            out = star_b.evaluate()
            gq = out['grad_moffat'].astype(np.float64).reshape(F, S, 4)
            gs = out['grad_stars'].astype(np.float64).reshape(F, S, 4)
            G = np.zeros((F, D))
            # chain rule through the (smooth) map theta -> q by central differences in double: the 13 + 13 perturbed parameter
            # sets go through quadratic_forms as ONE batch (27 calls on 800-element arrays per evaluation were two thirds of
            # the analytic stage of a C2-sized fit)
            hs = 1e-6 * np.maximum(1.0, np.abs(X[:, :13]).max(axis=0))             # [13]
            step = np.eye(13)[:, None, :] * hs[:, None, None]                       # [13][1][13]
            both = np.concatenate([X[None, :, :13] + step, X[None, :, :13] - step])  # [26][F][13]
            qb = quadratic_forms(both, xy, ss)                                       # [26][F][S][4]
            dq = (qb[:13] - qb[13:]) / (2.0 * hs[:, None, None, None])
            G[:, :13] = (gq[None] * dq).sum(axis=(2, 3)).T
            G[:, 13:13 + S], G[:, 13 + S:13 + 2 * S], G[:, 13 + 2 * S:] = gs[..., 0], gs[..., 1], gs[..., 2]
            np.ctypeslib.as_array(Fp, shape=(F,))[:] = out['loss'].astype(np.float64).reshape(F, S).sum(axis=1)
            np.ctypeslib.as_array(Gp, shape=(F, D))[:] = G
            return 0

        cb = _lib.LBFGS_EVAL(evaluate)
        dp = _lib.dp
        xf = np.ascontiguousarray(x)
        fl = np.zeros(F)
        nev = C.c_int()
        rc = lib.lc_batched_lbfgs(F, D, xf.ctypes.data_as(dp), np.ascontiguousarray(lo).ctypes.data_as(dp),
                                  np.ascontiguousarray(hi).ctypes.data_as(dp), n_iter_analytic, cb, None,
                                  fl.ctypes.data_as(dp), C.byref(nev))
        if failure:
            raise failure[0]
        if rc:
            raise _lib.LcError(f'lc_batched_lbfgs failed with status {rc}')
        push(xf)
        theta = xf[:, :13]
        # ---- stage B: free the pixel grid; l1-starlet weights from the noise propagation of the undistorted stars ------
        frame_b.set_moffat(theta[:, 0:4])
        frame_b.set_grid(None)
        st_now = star_b.get_stars().reshape(F, S, 4)
        tmp = PsfBatch(data, weight, ss, ctx)
        try:
            tmp.set_moffat(theta[:, 0:4])
            tmp.set_stars(st_now)
            tmp.propagate_noise()
            W = tmp.get_weights()
        finally:
            tmp.close()
        frame_b.set_regularization(W, lam_scales, lam_hf)
        frame_b.set_distortion(S, theta[:, 4:13], xy)
        star_b.set_regularization(None, 0.0, 0.0)
        cfg = dict(init_learning_rate=init_learning_rate, schedule_learning_rate=schedule_learning_rate)
        # per iteration: B resampled for every star -> step of the stars (chi2, its gradients, a, x0, y0) -> d chi2 / d B
        # summed over the stars by the adjoint resampling -> starlet term + step of B; the loop runs in the library
        frame_b.distortion_run(star_b, n_iter_adabelief, **cfg)
        hist = frame_b.loss_history() + star_b.loss_history().reshape(F, S, -1).sum(axis=1)
        res_f = frame_b.results()
        res_s = star_b.results()
        st = star_b.get_stars().astype(np.float64).reshape(F, S, 4)
        grid = frame_b.get_grid()
    finally:
        star_b.close()
        frame_b.close()

    out = []
    resid_all = res_s['residuals'].reshape(F, S, n, n)
    if pad:   # back to the caller's size (the ring carried no weight: chi2 counts the caller's pixels only)
        P, Nu = pad * ss, n_user * ss
        resid_all = resid_all[..., pad:pad + n_user, pad:pad + n_user]
        weight = weight[..., pad:pad + n_user, pad:pad + n_user]
        res_f = dict(res_f)
        for key in ('narrow_psf', 'full_psf'):
            cut = res_f[key][:, P:P + Nu, P:P + Nu]
            res_f[key] = cut / cut.sum(axis=(-1, -2), keepdims=True)
        grid = np.ascontiguousarray(grid[:, P:P + Nu, P:P + Nu])
        N = Nu
    for f in range(F):
        Sf = S_list[f]
        wf = weight[f, :Sf]
        chi2 = float((wf * resid_all[f, :Sf].astype(np.float64) ** 2).sum() / max((wf > 0).sum(), 1))
        resid = resid_all[f, :Sf].astype(np.float64) * norms[f]
        kwargs_psf = {
            'kwargs_moffat': {'fwhm_x': np.array([theta[f, 0]]), 'fwhm_y': np.array([theta[f, 1]]),
                              'phi': np.array([theta[f, 2]]), 'beta': np.array([theta[f, 3]]), 'C': np.array([1.0])},
            'kwargs_gaussian': {'a': st[f, :Sf, 0] * norms[f], 'x0': st[f, :Sf, 1], 'y0': st[f, :Sf, 2]},
            'kwargs_background': {'background': grid[f].reshape(N * N), 'mean': st[f, :Sf, 3] * norms[f]},
            'kwargs_distortion': {'dilation_x': theta[f, 4:7].copy(), 'dilation_y': theta[f, 7:10].copy(),
                                  'shear': theta[f, 10:13].copy()},
        }
        out.append({
            'full_psf': res_f['full_psf'][f], 'narrow_psf': res_f['narrow_psf'][f],
            'models': np.asarray(images[f], dtype=np.float64) - resid, 'residuals': resid, 'kwargs_psf': kwargs_psf,
            'chi2': chi2, 'norm': norms[f], 'analytic_extra_fields': {'final_loss': float(fl[f])},
            'adabelief_extra_fields': {'loss_history': hist[f, 1:].astype(np.float64).tolist(), 'initial_loss': float(hist[f, 0])},
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
