"""build_psf through the facade: the structural contract of the reference's
tests/test_starred_calls/test_starred_calls.py:66-80 on its own fixture (seeded here), and a
converged-fit comparison with the oracle run through the same two stages."""
import numpy as np
import pytest

from lightcurver_amd.synthetic import make_psf_dataset

pytestmark = pytest.mark.gpu


def _reference_fixture(seed=0):
    # fixture of the reference test (test_starred_calls.py:10-18), seeded
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(-8, 8), np.arange(-8, 8))
    gauss = np.exp(-0.1 * (x ** 2 + y ** 2))
    data = 0.1 * rng.random((5, 16, 16)) + np.repeat(gauss[None, :, :], repeats=5, axis=0)
    noisemap = 0.1 * np.ones((5, 16, 16))
    return data, noisemap


def test_build_psf_structural_contract():
    from lightcurver_amd.starred.procedures.psf_routines import build_psf
    data, noisemap = _reference_fixture()
    result = build_psf(data, noisemap, subsampling_factor=1, n_iter_analytic=5, n_iter_adabelief=10,
                       masks=np.ones_like(data), guess_method_star_position='center')
    assert isinstance(result, dict)
    for key in ('full_psf', 'adabelief_extra_fields', 'narrow_psf', 'chi2', 'residuals'):
        assert key in result
    assert len(result['adabelief_extra_fields']['loss_history']) == 10
    assert result['residuals'].shape == data.shape
    assert result['narrow_psf'].shape == (16, 16) and result['full_psf'].shape == (16, 16)
    assert abs(result['narrow_psf'].sum() - 1) < 1e-5 and abs(result['full_psf'].sum() - 1) < 1e-5
    assert f"{result['chi2']:.02f}"
    km = result['kwargs_psf']['kwargs_moffat']
    assert np.isfinite(float((0.5 * (km['fwhm_x'] + km['fwhm_y'])).item()))
    assert isinstance(result['kwargs_psf']['kwargs_distortion'], dict)


def test_build_psf_recovers_synthetic_psf():
    from lightcurver_amd.starred.procedures.psf_routines import build_psf_batch
    ds = make_psf_dataset(F=3, S=6, n=32, ss=2, seed=42)
    res = build_psf_batch(list(ds['data']), list(ds['noisemap']), 2, masks=list(ds['masks']),
                          n_iter_analytic=60, n_iter_adabelief=600, guess_method_star_position='center',
                          guess_fwhm_pixels=ds['fwhm_guess'])
    for f, r in enumerate(res):
        assert r['chi2'] < 2.0  # acceptance criterion of the reference's integration test
        truth = ds['truth']['narrow_psf'][f]
        # noise- and regularisation-limited (6 stars): the oracle reaches the same ~7 % of the peak
        assert np.abs(r['narrow_psf'] - truth).max() < 0.15 * truth.max()
        lh = np.array(r['adabelief_extra_fields']['loss_history'])
        assert len(lh) == 600 and np.all(np.isfinite(lh))
        # fluxes and positions recovered (noise-limited, so loose)
        a = r['kwargs_psf']['kwargs_gaussian']['a']
        bright = ds['truth']['flux'][f] > 5.0  # faint stars are noise dominated
        assert np.allclose(a[bright], ds['truth']['flux'][f][bright], rtol=0.05)
        dx0 = np.abs(r['kwargs_psf']['kwargs_gaussian']['x0'] - ds['truth']['x0'][f])
        assert dx0[bright].max() < 0.05


def test_ragged_star_counts():
    from lightcurver_amd.starred.procedures.psf_routines import build_psf_batch
    ds = make_psf_dataset(F=2, S=5, n=16, ss=2, seed=9)
    imgs = [ds['data'][0], ds['data'][1][:3]]
    nois = [ds['noisemap'][0], ds['noisemap'][1][:3]]
    res = build_psf_batch(imgs, nois, 2, n_iter_analytic=20, n_iter_adabelief=50,
                          guess_method_star_position='center', guess_fwhm_pixels=ds['fwhm_guess'])
    assert res[0]['residuals'].shape == (5, 16, 16) and res[1]['residuals'].shape == (3, 16, 16)
    single = build_psf_batch([imgs[1]], [nois[1]], 2, n_iter_analytic=20, n_iter_adabelief=50,
                             guess_method_star_position='center', guess_fwhm_pixels=ds['fwhm_guess'][1:2])
    # a frame padded with zero-weight stamps gives the same fit as the frame alone
    assert np.allclose(res[1]['narrow_psf'], single[0]['narrow_psf'], atol=1e-6)
