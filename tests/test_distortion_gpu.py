"""apply_distortion (kernel K13, csrc/distort.hip) through the STARRED-shaped facade against the float64 oracle
(oracle/model.py apply_distortion; scipy.ndimage.map_coordinates pins the oracle's interpolation in
tests/test_oracle_cpu.py), with the call shapes of the reference (star_photometry.py:291-304,
roi_file_preparation.py:169-180: kwargs read back key by key from the regions file, one rescaled position).
Tolerance: fp32 bilinear resampling vs float64, 2e-6 of the peak."""
import numpy as np
import pytest

from oracle import model as om
from lightcurver_amd.synthetic import make_narrow_psf

pytestmark = pytest.mark.gpu


def _psf(N, ss, seed):
    rng = np.random.default_rng(seed)
    return make_narrow_psf(rng, N, ss)[0]


@pytest.mark.parametrize('N,ss', [(32, 2), (48, 2), (64, 2), (33, 1)])
def test_apply_distortion_matches_oracle(ctx, N, ss):
    from lightcurver_amd.starred.psf.psf import apply_distortion, distortion_coefficients
    psf = _psf(N, ss, 5 + N)
    kw = {'dilation_x': np.array([0.01, 0.04, -0.02]), 'dilation_y': np.array([-0.015, 0.01, 0.03]),
          'shear': np.array([0.004, -0.02, 0.01])}
    coef = distortion_coefficients(kw)
    for xy in ([0.0, 0.0], [0.31, -0.42], [-0.5, 0.5]):
        got = apply_distortion(psf, kw, np.array([xy]), ctx=ctx)       # (1, 2) position as the reference passes it
        ref = om.apply_distortion(psf, coef, xy[0], xy[1]).numpy()
        assert got.shape == (N, N)
        assert np.abs(got - ref).max() < 2e-6 * ref.max()
        assert abs(got.sum() - 1.0) < 1e-5
    many = apply_distortion(psf, kw, np.array([[0.1, 0.2], [-0.3, 0.4], [0.5, -0.5]]), ctx=ctx)
    assert many.shape == (3, N, N)
    assert np.abs(many[1] - om.apply_distortion(psf, coef, -0.3, 0.4).numpy()).max() < 2e-6 * many[1].max()


def test_identity_and_errors(ctx):
    from lightcurver_amd import _lib
    from lightcurver_amd.starred.psf.psf import apply_distortion
    psf = _psf(32, 2, 1)
    same = apply_distortion(psf, {}, np.array([0.2, -0.1]), ctx=ctx)    # empty kwargs: no distortion
    assert np.abs(same - psf / psf.sum()).max() < 1e-7
    zero = apply_distortion(psf, {k: np.zeros(3) for k in ('dilation_x', 'dilation_y', 'shear')}, [0.2, -0.1], ctx=ctx)
    assert np.abs(zero - same).max() == 0.0
    with pytest.raises(KeyError):
        apply_distortion(psf, {'twist': np.zeros(3)}, [0.0, 0.0], ctx=ctx)
    with pytest.raises(_lib.LcError):   # inverting matrix
        apply_distortion(psf, {'dilation_x': np.array([-2.0, 0, 0])}, [0.0, 0.0], ctx=ctx)
