"""Float64 CPU restatement of the STARRED models lightcurver calls on its hot path.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py`` (parity unpinned).

Everything is written as a differentiable torch-float64 forward; gradients come
from ``torch.autograd`` exactly as the reference obtains them from JAX autodiff
of STARRED's forward model.  The hand-derived HIP gradients are checked against
these.

Reference call sites restated here (relative to /root/reference):
  * ``Deconv.model`` / ``setup_model``     - lightcurver/processes/star_photometry.py:66-69,124
                                             lightcurver/processes/roi_modelling.py:213-219,470
  * ``Loss`` (chi2 + l1_starlet + ...)     - star_photometry.py:95-111, roi_modelling.py:275-276,313-321
  * ``propagate_noise(method='SLIT')``     - star_photometry.py:108-110, roi_modelling.py:299-301
  * ``build_psf`` PSF model                - lightcurver/processes/psf_modelling.py:164-171
  * ``FisherCovariance(diagonal_only)``    - lightcurver/utilities/starred_utilities.py:36-38
Conventions (frozen in DESIGN.md "SPEC"):
  x = columns (axis -1), y = rows (axis -2); origin at the stamp centre; data-pixel units for
  c_x, c_y, dx, dy, x0, y0, fwhm; alpha in degrees (roi_modelling.py:53-55 de-rotates with
  scipy.ndimage.rotate, which takes degrees).
"""
import math

import numpy as np
import torch

DT = torch.float64
GAUSS_FWHM = 2.0  # target resolution, high-res pixels
SIGMA_G = GAUSS_FWHM / (2.0 * math.sqrt(2.0 * math.log(2.0)))
B3 = (1.0 / 16, 4.0 / 16, 6.0 / 16, 4.0 / 16, 1.0 / 16)


def T(x, dtype=DT):
    if torch.is_tensor(x):
        return x.to(dtype)
    return torch.as_tensor(np.asarray(x, dtype=np.float64), dtype=dtype)


def cref(N):
    """Zero-lag index of an N-sample kernel under scipy's mode='same' cropping."""
    return (N - 1) // 2


def n_scales(N):
    return int(math.log2(N))


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
def gaussian_stack(N, X, Y, amp, sigma=SIGMA_G):
    """sum_i amp_i * G(u, v; X_i, Y_i) on an N x N grid.  X, Y, amp: (..., M) -> (..., N, N).

    G is the continuous unit-integral Gaussian sampled at pixel centres
    (1 / (2 pi sigma^2)) exp(-((v-X)^2 + (u-Y)^2) / (2 sigma^2)), evaluated on the full grid.
    """
    v = torch.arange(N, dtype=X.dtype)
    gx = torch.exp(-0.5 * ((v - X[..., None]) / sigma) ** 2)
    gy = torch.exp(-0.5 * ((v - Y[..., None]) / sigma) ** 2)
    norm = amp / (2.0 * math.pi * sigma * sigma)
    return torch.einsum('...m,...mu,...mv->...uv', norm, gy, gx)


def conv_same(a, k):
    """scipy.signal.fftconvolve(a, k, mode='same') over the last two axes (zero padded, linear)."""
    n1, n2 = a.shape[-2:]
    k1, k2 = k.shape[-2:]
    l1, l2 = n1 + k1 - 1, n2 + k2 - 1
    f = torch.fft.rfft2(a, s=(l1, l2)) * torch.fft.rfft2(k, s=(l1, l2))
    full = torch.fft.irfft2(f, s=(l1, l2))
    s1, s2 = (k1 - 1) // 2, (k2 - 1) // 2
    return full[..., s1:s1 + n1, s2:s2 + n2]


def blocksum(img, ss):
    """Flux-conserving down-sampling: sum over ss x ss blocks of the last two axes."""
    if ss == 1:
        return img
    *lead, n1, n2 = img.shape
    return img.reshape(*lead, n1 // ss, ss, n2 // ss, ss).sum(dim=(-1, -3))


def upsample_rep(img, ss):
    """Adjoint of blocksum: replicate every data pixel over its ss x ss high-res block."""
    if ss == 1:
        return img
    return img.repeat_interleave(ss, dim=-2).repeat_interleave(ss, dim=-1)


def bilinear_clamp(h, Ys, Xs):
    """Order-1 interpolation of h (N, N) at fractional index coords, edge-replicating outside.

    Same as scipy.ndimage.map_coordinates(h, [Ys, Xs], order=1, mode='nearest').
    """
    N = h.shape[-1]
    x0 = torch.floor(Xs)
    y0 = torch.floor(Ys)
    fx = Xs - x0
    fy = Ys - y0
    x0 = x0.long()
    y0 = y0.long()
    xa, xb = x0.clamp(0, N - 1), (x0 + 1).clamp(0, N - 1)
    ya, yb = y0.clamp(0, N - 1), (y0 + 1).clamp(0, N - 1)
    top = (1 - fx) * h[ya, xa] + fx * h[ya, xb]
    bot = (1 - fx) * h[yb, xa] + fx * h[yb, xb]
    return (1 - fy) * top + fy * bot


def starlet(img, J):
    """Isotropic undecimated (a trous) B3-spline starlet, first generation, edge-replicating.

    Returns (J + 1, ..., N, N): J detail scales (finest first) then the coarse residual.
    """
    N1, N2 = img.shape[-2:]
    i1 = torch.arange(N1)
    i2 = torch.arange(N2)
    c = img
    out = []
    for j in range(J):
        d = 2 ** j
        r = sum(B3[t + 2] * c[..., (i1 + t * d).clamp(0, N1 - 1), :] for t in range(-2, 3))
        cn = sum(B3[t + 2] * r[..., :, (i2 + t * d).clamp(0, N2 - 1)] for t in range(-2, 3))
        out.append(c - cn)
        c = cn
    out.append(c)
    return torch.stack(out)


def starlet_norms(N, J, dtype=DT):
    """l2 norm of the starlet atom of every scale (transform of a dirac at the zero-lag index,
    on the N x N grid itself, edge effects included)."""
    d = torch.zeros(N, N, dtype=dtype)
    d[cref(N), cref(N)] = 1.0
    return torch.sqrt((starlet(d, J) ** 2).sum(dim=(-1, -2)))


def l1_starlet(img, W, lam_scales, lam_hf, J):
    """lam_hf * sum W_0 |w_0| + lam_scales * sum_{1<=j<J} W_j |w_j|  (coarse scale not penalised)."""
    st = starlet(img, J)
    hf = (W[0] * st[0].abs()).sum()
    rest = (W[1:J] * st[1:J].abs()).sum() if J > 1 else 0.0
    return lam_hf * hf + lam_scales * rest


def default_W(N, J, dtype=DT):
    """Weights used when no noise propagation is supplied: the starlet scale norms."""
    return starlet_norms(N, J, dtype)[:, None, None] * torch.ones(J + 1, N, N, dtype=dtype)


# ----------------------------------------------------------------------------------------------
# Deconv (joint multi-epoch forward model)
# ----------------------------------------------------------------------------------------------
def deconv_scene(p, E, M, N, ss):
    """High-resolution scene of every epoch before convolution: point sources + resampled h."""
    c0 = (N - 1) / 2.0
    al = p['alpha'] * (math.pi / 180.0)
    ca, sa = torch.cos(al)[:, None], torch.sin(al)[:, None]
    cx, cy = p['c_x'][None, :], p['c_y'][None, :]
    X = c0 + ss * (ca * cx - sa * cy + p['dx'][:, None])
    Y = c0 + ss * (sa * cx + ca * cy + p['dy'][:, None])
    P = gaussian_stack(N, X, Y, p['a'].reshape(E, M))
    idx = torch.arange(N, dtype=p['h'].dtype)
    px = (idx - c0)[None, None, :] - ss * p['dx'][:, None, None]
    py = (idx - c0)[None, :, None] - ss * p['dy'][:, None, None]
    ca3, sa3 = ca[:, :, None], sa[:, :, None]
    Xs = c0 + ca3 * px + sa3 * py
    Ys = c0 - sa3 * px + ca3 * py
    Hs = bilinear_clamp(p['h'].reshape(N, N), Ys, Xs)
    return P, Hs


def deconv_model(p, psf, ss, n):
    """Deconv.model(kwargs) -> (E, n, n).  p: dict a (E*M, epoch-major), c_x, c_y (M), dx, dy,
    alpha, mean (E), h (N*N)."""
    E = psf.shape[0]
    N = ss * n
    M = p['c_x'].shape[0]
    P, Hs = deconv_scene(p, E, M, N, ss)
    conv = conv_same(P + Hs, psf)
    return blocksum(conv, ss) + p['mean'][:, None, None]


def deconv_deconvolved(p, epoch, N, ss):
    """Deconv.getDeconvolved(kwargs, epoch) -> (scene incl. point sources, background only)."""
    E = p['dx'].shape[0]
    M = p['c_x'].shape[0]
    P, Hs = deconv_scene(p, E, M, N, ss)
    return P[epoch] + Hs[epoch], Hs[epoch]


def deconv_loss(p, data, sigma2, psf, ss, W=None, lam_scales=0.0, lam_hf=0.0, lam_pos=0.0,
                lam_pos_ps=0.0, lam_pts=0.0, lam_fu=0.0, prior=None, parts=False):
    """Loss of STARRED's deconvolution as frozen in DESIGN.md:

    0.5*chi2 + l1_starlet(h) + positivity(h) + positivity(a) + pts_source + flux_uniformity + prior
    prior: iterable of (name, mean, sigma).
    """
    E, n, _ = data.shape
    N = ss * n
    M = p['c_x'].shape[0]
    J = n_scales(N)
    m = deconv_model(p, psf, ss, n)
    chi2 = ((data - m) ** 2 / sigma2).sum()
    terms = {'chi2': 0.5 * chi2}
    h = p['h'].reshape(N, N)
    if W is None:
        W = default_W(N, J, h.dtype)
    if lam_scales != 0.0 or lam_hf != 0.0:
        terms['l1'] = l1_starlet(h, W, lam_scales, lam_hf, J)
    if lam_pos != 0.0:
        terms['pos'] = lam_pos * torch.clamp(-h, min=0.0).sum()
    if lam_pos_ps != 0.0:
        terms['pos_ps'] = lam_pos_ps * torch.clamp(-p['a'], min=0.0).sum()
    if lam_pts != 0.0:
        c0 = (N - 1) / 2.0
        abar = p['a'].reshape(E, M).mean(dim=0)
        pbar = gaussian_stack(N, c0 + ss * p['c_x'], c0 + ss * p['c_y'], abar)
        terms['pts'] = lam_pts * (W[0] * starlet(pbar, J)[0].abs()).sum()
    if lam_fu != 0.0 and E > 1:
        a2 = p['a'].reshape(E, M)
        var = ((a2 - a2.mean(dim=0, keepdim=True)) ** 2).mean(dim=0)
        terms['fu'] = lam_fu * torch.sqrt(var).sum()
    if prior:
        terms['prior'] = sum((((p[name] - T(mu)) / T(sg)) ** 2).sum() * 0.5 for name, mu, sg in prior)
    total = sum(terms.values())
    return (total, terms) if parts else total


def _shift_zero(img, sh):
    """out[..., t] = img[..., t + sh] along the last two axes, zero filled."""
    out = torch.zeros_like(img)
    N = img.shape[-1]
    lo, hi = max(0, -sh), min(N, N - sh)
    if hi > lo:
        out[..., lo:hi, lo:hi] = img[..., lo + sh:hi + sh, lo + sh:hi + sh]
    return out


def noise_levels_from_impulse(r, w, ss):
    """Standard deviation of the chi2-gradient noise in every starlet scale.

    r: (K, N, N) response of the gradient image to a unit of (Sigma^-1 noise) in the central data pixel
    p* = (n//2, n//2) of contributor k (an epoch, or a star of a frame); w: (K, n, n) its inverse variances.
    With kappa_{k,j} = starlet(r_k)[j] and shift invariance, the coefficient at x responds to data pixel p
    with kappa_{k,j}(x - ss p + ss p*), hence
        W_j(x)^2 = sum_k sum_p w_k[p] kappa_{k,j}(x - ss p + ss p*)^2 = sum_k ( up0(w_k) (*) kappa_{k,j}^2 )(x)
    (up0 = zero-insertion up-sampling).  Returns (J + 1, N, N).
    """
    K, N, _ = r.shape
    n = N // ss
    J = n_scales(N)
    kap = starlet(r, J)  # (J + 1, K, N, N)
    kern = _shift_zero(kap, ss * (n // 2) - cref(N)) ** 2
    up0 = torch.zeros(K, N, N, dtype=r.dtype)
    up0[:, ::ss, ::ss] = w
    lev = conv_same(up0[None].expand(J + 1, K, N, N), kern).sum(dim=1)
    return torch.sqrt(torch.clamp(lev, min=0.0))


def propagate_noise_deconv(sigma2, psf, ss):
    """propagate_noise(model, noisemap, ..., method='SLIT', likelihood_type='chi2')[0]:
    noise level of d(chi2/2)/dh per starlet scale, epochs added in quadrature (see
    noise_levels_from_impulse).  Returns (J + 1, N, N)."""
    E, n, _ = sigma2.shape
    N = ss * n
    x = torch.zeros(E, N, N, dtype=psf.dtype, requires_grad=True)
    y = blocksum(conv_same(x, psf), ss)[:, n // 2, n // 2].sum()
    (r,) = torch.autograd.grad(y, x)
    w = torch.where(torch.isfinite(sigma2) & (sigma2 > 0), 1.0 / sigma2, torch.zeros_like(sigma2))
    return noise_levels_from_impulse(r, w, ss)


def fisher_flux_sigma(p, sigma2, psf, ss):
    """FisherCovariance(diagonal_only=True) with only `a` free: 1/sqrt(sum (dm/da)^2 / sigma^2)."""
    E, n, _ = sigma2.shape
    N = ss * n
    M = p['c_x'].shape[0]
    c0 = (N - 1) / 2.0
    al = p['alpha'] * (math.pi / 180.0)
    ca, sa = torch.cos(al)[:, None], torch.sin(al)[:, None]
    X = c0 + ss * (ca * p['c_x'][None] - sa * p['c_y'][None] + p['dx'][:, None])
    Y = c0 + ss * (sa * p['c_x'][None] + ca * p['c_y'][None] + p['dy'][:, None])
    out = torch.zeros(E, M, dtype=psf.dtype)
    for i in range(M):
        g = gaussian_stack(N, X[:, i:i + 1], Y[:, i:i + 1], torch.ones(E, 1, dtype=psf.dtype))
        d = blocksum(conv_same(g, psf), ss)
        out[:, i] = 1.0 / torch.sqrt((d ** 2 / sigma2).sum(dim=(-1, -2)))
    return out.reshape(E * M)


# ----------------------------------------------------------------------------------------------
# PSF model (build_psf)
# ----------------------------------------------------------------------------------------------
def moffat(N, ss, fwhm_x, fwhm_y, phi, beta):
    """Elliptical Moffat, unit sum on the N x N grid, centred on the zero-lag index cref(N).
    fwhm in data pixels, phi in radians."""
    c = cref(N)
    idx = torch.arange(N, dtype=fwhm_x.dtype) - c
    x, y = idx[None, :], idx[:, None]
    k = 2.0 * torch.sqrt(2.0 ** (1.0 / beta) - 1.0)
    ax, ay = ss * fwhm_x / k, ss * fwhm_y / k
    xr = x * torch.cos(phi) + y * torch.sin(phi)
    yr = -x * torch.sin(phi) + y * torch.cos(phi)
    m = (1.0 + (xr / ax) ** 2 + (yr / ay) ** 2) ** (-beta)
    return m / m.sum()


def psf_narrow_unnormalised(p, N, ss):
    return moffat(N, ss, p['fwhm_x'], p['fwhm_y'], p['phi'], p['beta']) + p['B'].reshape(N, N)


def psf_model(p, ss, n):
    """PSF.model(kwargs) -> (S, n, n): a_i * D[ G(x0_i, y0_i) (*) (Moffat + B) ] + sky_i."""
    N = ss * n
    c0 = (N - 1) / 2.0
    Tn = psf_narrow_unnormalised(p, N, ss)
    S = p['a'].shape[0]
    X = (c0 + ss * p['x0'])[:, None]
    Y = (c0 + ss * p['y0'])[:, None]
    G = gaussian_stack(N, X, Y, torch.ones(S, 1, dtype=Tn.dtype))
    conv = conv_same(G, Tn[None].expand(S, N, N))
    return p['a'][:, None, None] * blocksum(conv, ss) + p['sky'][:, None, None]


def psf_loss(p, data, sigma2, mask, ss, W=None, lam_scales=0.0, lam_hf=0.0, parts=False):
    S, n, _ = data.shape
    N = ss * n
    J = n_scales(N)
    m = psf_model(p, ss, n)
    chi2 = (mask * (data - m) ** 2 / sigma2).sum()
    terms = {'chi2': 0.5 * chi2}
    if lam_scales != 0.0 or lam_hf != 0.0:
        if W is None:
            W = default_W(N, J, m.dtype)
        terms['l1'] = l1_starlet(p['B'].reshape(N, N), W, lam_scales, lam_hf, J)
    total = sum(terms.values())
    return (total, terms) if parts else total


def psf_outputs(p, ss, n):
    """narrow_psf (unit sum), full_psf = narrow (*) G(centre) (unit sum)."""
    N = ss * n
    c0 = (N - 1) / 2.0
    Tn = psf_narrow_unnormalised(p, N, ss)
    narrow = Tn / Tn.sum()
    g = gaussian_stack(N, torch.tensor([c0], dtype=Tn.dtype), torch.tensor([c0], dtype=Tn.dtype),
                       torch.ones(1, dtype=Tn.dtype))
    full = conv_same(g, narrow)
    return narrow, full / full.sum()


def propagate_noise_psf(p, sigma2, mask, ss):
    """Noise level of d(chi2/2)/dB per starlet scale (J + 1, N, N): the stars of the frame add in
    quadrature, each through the impulse response of its own shifted Gaussian (times its amplitude)."""
    S, n, _ = sigma2.shape
    N = ss * n
    c0 = (N - 1) / 2.0
    X = (c0 + ss * p['x0'])[:, None]
    Y = (c0 + ss * p['y0'])[:, None]
    G = gaussian_stack(N, X, Y, torch.ones(S, 1, dtype=sigma2.dtype))
    x = torch.zeros(S, N, N, dtype=sigma2.dtype, requires_grad=True)
    y = (p['a'][:, None, None] * blocksum(conv_same(G, x), ss))[:, n // 2, n // 2].sum()
    (r,) = torch.autograd.grad(y, x)
    return noise_levels_from_impulse(r, mask / sigma2, ss)


def distortion_matrix(coef, x, y):
    """A = [[1 + dilation_x, shear], [shear, 1 + dilation_y]], each entry c0 + c1 x + c2 y (coef: 9 values in the order
    dilation_x, dilation_y, shear).  DESIGN.md section 3 (frozen, unverified against STARRED)."""
    c = T(coef).reshape(3, 3)
    v = c[:, 0] + c[:, 1] * x + c[:, 2] * y
    return torch.stack([torch.stack([1.0 + v[0], v[2]]), torch.stack([v[2], 1.0 + v[1]])])


def apply_distortion(narrow_psf, coef, x, y):
    """starred.psf.psf.apply_distortion(narrow_psf, kwargs_distortion, star_xy_coordinates) as frozen in DESIGN.md:
    out[u, v] = bilinear_0(psf, c + A^-1 ((v, u) - c)), c = cref(N), zeros outside the grid, unit sum.
    Reference call sites: lightcurver/processes/star_photometry.py:303-304, roi_file_preparation.py:179-180."""
    N = narrow_psf.shape[-1]
    c = float(cref(N))
    Ai = torch.linalg.inv(distortion_matrix(coef, T(x), T(y)))
    idx = torch.arange(N, dtype=DT) - c
    qx, qy = idx[None, :], idx[:, None]
    X = c + Ai[0, 0] * qx + Ai[0, 1] * qy
    Y = c + Ai[1, 0] * qx + Ai[1, 1] * qy
    x0, y0 = torch.floor(X), torch.floor(Y)
    fx, fy = X - x0, Y - y0
    x0, y0 = x0.long(), y0.long()
    p = T(narrow_psf)

    def at(yy, xx):
        ok = (yy >= 0) & (yy < N) & (xx >= 0) & (xx < N)
        return torch.where(ok, p[yy.clamp(0, N - 1), xx.clamp(0, N - 1)], torch.zeros((), dtype=DT))

    top = (1 - fx) * at(y0, x0) + fx * at(y0, x0 + 1)
    bot = (1 - fx) * at(y0 + 1, x0) + fx * at(y0 + 1, x0 + 1)
    out = (1 - fy) * top + fy * bot
    return out / out.sum()


def moffat_distorted(N, ss, fwhm_x, fwhm_y, phi, beta, A):
    """The elliptical Moffat seen through the distortion A: M(A^-1 (u - c)), unit sum on the grid."""
    c = cref(N)
    idx = torch.arange(N, dtype=fwhm_x.dtype) - c
    x, y = idx[None, :], idx[:, None]
    Ai = torch.linalg.inv(A)
    qx, qy = Ai[0, 0] * x + Ai[0, 1] * y, Ai[1, 0] * x + Ai[1, 1] * y
    k = 2.0 * torch.sqrt(2.0 ** (1.0 / beta) - 1.0)
    ax, ay = ss * fwhm_x / k, ss * fwhm_y / k
    xr = qx * torch.cos(phi) + qy * torch.sin(phi)
    yr = -qx * torch.sin(phi) + qy * torch.cos(phi)
    m = (1.0 + (xr / ax) ** 2 + (yr / ay) ** 2) ** (-beta)
    return m / m.sum()


def warp_grid(B, A):
    """W_A[B](u) = bilinear_0(B, c + A^-1 (u - c)) / det A: the pixel grid resampled through the distortion."""
    N = B.shape[-1]
    c = float(cref(N))
    Ai = torch.linalg.inv(A)
    idx = torch.arange(N, dtype=DT) - c
    qx, qy = idx[None, :], idx[:, None]
    X = c + Ai[0, 0] * qx + Ai[0, 1] * qy
    Y = c + Ai[1, 0] * qx + Ai[1, 1] * qy
    x0, y0 = torch.floor(X).detach(), torch.floor(Y).detach()
    fx, fy = X - x0, Y - y0
    x0, y0 = x0.long(), y0.long()

    def at(yy, xx):
        ok = (yy >= 0) & (yy < N) & (xx >= 0) & (xx < N)
        return torch.where(ok, B[yy.clamp(0, N - 1), xx.clamp(0, N - 1)], torch.zeros((), dtype=DT))

    top = (1 - fx) * at(y0, x0) + fx * at(y0, x0 + 1)
    bot = (1 - fx) * at(y0 + 1, x0) + fx * at(y0 + 1, x0 + 1)
    return ((1 - fy) * top + fy * bot) / torch.linalg.det(A)


def psf_model_distorted(p, xy, ss, n):
    """build_psf(field_distortion=True) as frozen in DESIGN.md section 3: star i at the rescaled frame coordinates
    xy[i] sees T_i = Moffat seen through A_i + W_{A_i}[B];  p['dist']: the 9 distortion coefficients."""
    N = ss * n
    c0 = (N - 1) / 2.0
    S = p['a'].shape[0]
    B = p['B'].reshape(N, N)
    out = []
    for i in range(S):
        A = distortion_matrix(p['dist'], T(xy[i][0]), T(xy[i][1]))
        Ti = moffat_distorted(N, ss, p['fwhm_x'], p['fwhm_y'], p['phi'], p['beta'], A) + warp_grid(B, A)
        G = gaussian_stack(N, (c0 + ss * p['x0'][i]).reshape(1), (c0 + ss * p['y0'][i]).reshape(1), torch.ones(1, dtype=DT))
        out.append(p['a'][i] * blocksum(conv_same(G, Ti), ss) + p['sky'][i])
    return torch.stack(out)


def psf_loss_distorted(p, xy, data, sigma2, mask, ss, W=None, lam_scales=0.0, lam_hf=0.0):
    S, n, _ = data.shape
    N = ss * n
    J = n_scales(N)
    m = psf_model_distorted(p, xy, ss, n)
    total = 0.5 * (mask * (data - m) ** 2 / sigma2).sum()
    if lam_scales != 0.0 or lam_hf != 0.0:
        if W is None:
            W = default_W(N, J, m.dtype)
        total = total + l1_starlet(p['B'].reshape(N, N), W, lam_scales, lam_hf, J)
    return total


def reduced_chi2(data, model, sigma2, mask):
    return float((mask * (data - model) ** 2 / sigma2).sum() / mask.sum())
