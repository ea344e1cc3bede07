"""``Deconv`` / ``setup_model``: the joint multi-epoch forward model STARRED exposes and
lightcurver drives (reference call sites: lightcurver/processes/star_photometry.py:66-69,124,137;
roi_modelling.py:213-219,387,470).  The arithmetic runs in liblcmi.so (lc_joint_*)."""
import hashlib

import numpy as np

from ...joint import JointFit, make_joint_fit

ANALYTIC = ('a', 'c_x', 'c_y', 'dx', 'dy', 'alpha')
BACKGROUND = ('h', 'mean')


def flatten_kwargs(kwargs):
    """STARRED nested kwargs -> flat dict of float64 arrays (a, c_x, c_y, dx, dy, alpha, h, mean)."""
    out = {}
    for k in ANALYTIC:
        if k in kwargs.get('kwargs_analytic', {}):
            out[k] = np.atleast_1d(np.asarray(kwargs['kwargs_analytic'][k], dtype=np.float64)).ravel()
    for k in BACKGROUND:
        if k in kwargs.get('kwargs_background', {}):
            out[k] = np.atleast_1d(np.asarray(kwargs['kwargs_background'][k], dtype=np.float64)).ravel()
    return out


def nest_kwargs(flat):
    return {'kwargs_analytic': {k: np.array(flat[k], dtype=np.float32) for k in ANALYTIC if k in flat},
            'kwargs_background': {k: np.array(flat[k], dtype=np.float32) for k in BACKGROUND if k in flat},
            'kwargs_sersic': {}}


class Deconv:
    """Image deconvolution model: M point sources (flux per epoch, shared positions, per-epoch shifts
    and fixed rotation) + pixelated background h, convolved with each epoch's narrow PSF."""

    def __init__(self, image_size, number_of_sources, scale=1.0, upsampling_factor=2, epochs=1, psf=None,
                 gaussian_fwhm=2, convolution_method='fft', ctx=None):
        if gaussian_fwhm != 2:
            raise NotImplementedError('the target resolution is fixed to FWHM = 2 high-resolution pixels')
        self.image_size = int(image_size)
        self.upsampling_factor = int(upsampling_factor)
        self.image_size_up = self.image_size * self.upsampling_factor
        self.epochs = int(epochs)
        self.M = int(number_of_sources)
        self.scale = scale
        self.psf = None if psf is None else np.ascontiguousarray(psf, dtype=np.float32)
        self._ctx = ctx
        self._fit = None
        self._fit_key = None
        self._sigma2_f32 = None

    # -- device object management -------------------------------------------------------------------
    def _ensure_fit(self, data=None, sigma_2=None):
        """(Re)create the device object when the data / variance it was built on change."""
        if data is None:
            if self._fit is None:
                raise RuntimeError('this Deconv has no data attached yet: build it with setup_model or a Loss')
            return self._fit
        data = np.asarray(data)
        sigma_2 = np.asarray(sigma_2)
        # every byte counts: the reference masks / boosts single pixels between calls (star_photometry.py:309-316)
        key = (data.shape, str(data.dtype), _fingerprint(data), _fingerprint(sigma_2))
        if self._fit is None or key != self._fit_key:
            if self._fit is not None:
                self._fit.close()
            self._fit = make_joint_fit(data, sigma_2, self.psf, self.upsampling_factor, self.M, self._ctx)
            self._fit_key = key
            self._sigma2_f32 = np.asarray(sigma_2, dtype=np.float32).copy()  # what propagate_noise compares with
        return self._fit

    def _push(self, kwargs):
        fit = self._ensure_fit()
        fit.set_params(**flatten_kwargs(kwargs))
        return fit

    # -- STARRED API ---------------------------------------------------------------------------------
    def model(self, kwargs):
        """Modelled data cube (E, n, n) for the given kwargs."""
        fit = self._push(kwargs)
        return fit.model()[0]

    def getDeconvolved(self, kwargs, epoch=0):
        """(high-resolution scene incl. point sources at the target resolution, background only)."""
        fit = self._push(kwargs)
        return fit.deconvolved(epoch)


def _fingerprint(a):
    """128-bit fingerprint of an array's bytes, to notice that the caller changed the data between two calls.  xxh3 where the
    module is there (it reads the buffer in place at memory speed: 32 MB of float64 cubes cost 65 ms per Loss with blake2b
    over a copy - a tenth of a whole C4-sized two-stage fit), blake2b otherwise."""
    buf = memoryview(np.ascontiguousarray(a)).cast('B')
    try:
        import xxhash
        return xxhash.xxh3_128_digest(buf)
    except ImportError:
        return hashlib.blake2b(buf, digest_size=16).digest()


def setup_model(data, sigma_2, s, xs, ys, subsampling_factor, initial_a, ctx=None):
    """Build the model and the initial / bound / fixed kwargs, as STARRED's ``setup_model`` does.

    data, sigma_2: (E, n, n); s: (E, N, N) narrow PSFs; xs, ys: M initial positions (data pixels,
    origin at the stamp centre); initial_a: E*M fluxes, epoch-major.
    """
    data = np.asarray(data)
    E, n, _ = data.shape
    xs = np.atleast_1d(np.asarray(xs, dtype=np.float64))
    ys = np.atleast_1d(np.asarray(ys, dtype=np.float64))
    M = xs.size
    ss = int(subsampling_factor)
    N = n * ss
    a0 = np.atleast_1d(np.asarray(initial_a, dtype=np.float64)).ravel()
    if a0.size != E * M:
        raise ValueError(f'initial_a must have epochs * sources = {E * M} entries')
    model = Deconv(image_size=n, number_of_sources=M, scale=1.0, upsampling_factor=ss, epochs=E, psf=s, ctx=ctx)
    model._ensure_fit(data, sigma_2)
    init = dict(a=a0, c_x=xs, c_y=ys, dx=np.zeros(E), dy=np.zeros(E), alpha=np.zeros(E), h=np.zeros(N * N),
                mean=np.zeros(E))
    big = 1e10
    half = n / 2.0
    up = dict(a=np.full(E * M, big), c_x=xs + half, c_y=ys + half, dx=np.full(E, half), dy=np.full(E, half),
              alpha=np.full(E, 360.0), h=np.full(N * N, big), mean=np.full(E, big))
    down = dict(a=np.zeros(E * M), c_x=xs - half, c_y=ys - half, dx=np.full(E, -half), dy=np.full(E, -half),
                alpha=np.full(E, -360.0), h=np.full(N * N, -big), mean=np.full(E, -big))
    kwargs_init, kwargs_up, kwargs_down = nest_kwargs(init), nest_kwargs(up), nest_kwargs(down)
    kwargs_fixed = {'kwargs_analytic': {}, 'kwargs_background': {}, 'kwargs_sersic': {}}
    return model, kwargs_init, kwargs_up, kwargs_down, kwargs_fixed
