"""GPU parity of the fused stamp pre-processing (lc_prepare_stamps) against the NumPy
restatement of the reference lines (oracle/prep.py).  Tolerance: the kernel works in fp32, the oracle in fp64:
2e-6 relative on noise maps and weights; counts and masks exact; cleaned data exact to fp32 rounding of the
division (1e-7)."""
import numpy as np
import pytest

from oracle import prep as op

pytestmark = pytest.mark.gpu


def _stack(K, n, seed, nan_frac=0.01, bad_frac=0.02):
    rng = np.random.default_rng(seed)
    data = (rng.normal(0, 5, (K, n, n)) + 200 * np.exp(-((np.indices((n, n)) - n / 2) ** 2).sum(0) / 8)).astype(np.float32)
    rms = rng.uniform(2, 6, K).astype(np.float32)
    t = rng.uniform(30, 300, K).astype(np.float32)
    bad = rng.random((K, n, n)) < bad_frac
    nan = rng.random((K, n, n)) < nan_frac
    data[nan] = np.nan
    coef = rng.uniform(0.5, 2.0, K).astype(np.float32)
    return data, rms, t, bad, coef


def _close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(b)
    assert np.abs(a[m] - b[m]).max() <= tol * max(np.abs(b[m]).max(), 1e-300)


@pytest.mark.parametrize('K,n', [(1, 16), (7, 32), (40, 64), (3, 128), (5, 24)])
def test_noise_map_from_rms_and_psf_style_cleanup(ctx, K, n):
    """cutout_making.py:43-51 + psf_modelling.py:135-153: noise map from the rms, NaN pixels -> (0, 1), weights,
    masked-pixel count."""
    from lightcurver_amd.processes.preprocessing import prepare_stamps
    data, rms, t, bad, coef = _stack(K, n, 10 + K)
    out = prepare_stamps(data, rms=rms, exptime=t, bad=bad, nan_noise=1.0, ctx=ctx)
    d, s, w, cnt = op.prepare(data, rms=rms, exptime=t, bad=bad, nan_noise=1.0)
    np.testing.assert_array_equal(out['masked_count'], cnt)
    _close(out['data'], d, 1e-7)
    _close(out['noisemap'], s, 2e-6)
    _close(out['weight'], w, 4e-6)
    assert np.array_equal(out['weight'] > 0, w > 0)


@pytest.mark.parametrize('whole', [False, True])
def test_roi_style_cleanup(ctx, whole):
    """roi_file_preparation.py:162-201 (per-pixel boost) and star_photometry.py:309-316 (whole-epoch boost, once)."""
    from lightcurver_amd.processes.preprocessing import prepare_stamps
    K, n = 12, 32
    data, rms, t, bad, coef = _stack(K, n, 77)
    noise = op.noisemap_from_rms(data, rms, t).astype(np.float32)
    bad[3] = False  # one clean epoch
    bad[5] = True   # one fully flagged epoch
    out = prepare_stamps(data, noisemap=noise, coefficient=coef, bad=bad, nan_noise=1e7, noise_boost=1000.0,
                         boost_whole_stamp=whole, ctx=ctx)
    d, s, w, cnt = op.prepare(data, noisemap=noise, coefficient=coef, bad=bad, nan_noise=1e7, noise_boost=1000.0,
                              boost_whole_stamp=whole)
    np.testing.assert_array_equal(out['masked_count'], cnt)
    assert out['masked_count'][5] == n * n
    _close(out['data'], d, 2e-7)
    _close(out['noisemap'], s, 2e-6)
    _close(out['weight'], w, 4e-6)


def test_edge_cases(ctx):
    from lightcurver_amd.processes.preprocessing import prepare_stamps
    from lightcurver_amd._lib import LcError
    # NaN in the data only (noise map given and finite): stays NaN in data, weight 0, not counted as masked
    data = np.ones((2, 16, 16), np.float32)
    noise = np.full((2, 16, 16), 2.0, np.float32)
    data[0, 3, 4] = np.nan
    noise[1, 0, 0] = 0.0  # zero noise: weight 0 instead of inf
    out = prepare_stamps(data, noisemap=noise, ctx=ctx)
    assert np.isnan(out['data'][0, 3, 4]) and out['weight'][0, 3, 4] == 0 and out['weight'][1, 0, 0] == 0
    assert out['masked_count'].tolist() == [0, 0]
    assert np.all(out['weight'][0, 0] == 0.25)
    # clamp of the noise map at 1e-7 electrons (cutout_making.py:47)
    z = prepare_stamps(np.zeros((1, 16, 16), np.float32), rms=np.zeros(1), exptime=np.full(1, 10.0), ctx=ctx)
    assert np.allclose(z['noisemap'], 1e-8, rtol=1e-6)
    with pytest.raises(LcError):
        prepare_stamps(data, ctx=ctx)  # neither a noise map nor rms / exptime


def test_copy_bandwidth_probe(ctx):
    import ctypes as C
    from lightcurver_amd import _lib
    g = C.c_float()
    ctx.check(_lib.lib().lc_copy_bandwidth(ctx.h, 1 << 28, 5, C.byref(g)), 'lc_copy_bandwidth')
    assert 500.0 < g.value < 8000.0  # GB/s, read + write


def test_batched_psf_stamp_preparation_matches_per_frame_host_logic(ctx):
    """model_psfs_of_frames cleans the stamps of every frame in one device launch; the result must be what the
    per-frame host restatement of psf_modelling.py:135-153 gives, including the 40 % cut and a frame that
    loses all its stamps (skipped, psf_modelling.py:154-160)."""
    from lightcurver_amd.processes.psf_modelling import (model_psfs_of_frames, prepare_psf_stamps,
                                                         prepare_psf_stamps_batched)
    from lightcurver_amd.synthetic import make_psf_dataset
    ds = make_psf_dataset(F=3, S=4, n=32, ss=2, seed=31)
    rng = np.random.default_rng(5)
    frames = []
    for f in range(3):
        d = ds['data'][f].astype(np.float32).copy()
        nm = ds['noisemap'][f].astype(np.float32).copy()
        cosm = rng.random(d.shape) < 0.02
        d[0, 2, 3] = np.nan
        nm[0, 2, 3] = np.nan
        if f == 1:
            cosm[2] = rng.random(d.shape[1:]) < 0.6   # one stamp over the 40 % cut
        if f == 2:
            cosm[:] = True                            # every stamp rejected
        frames.append(dict(datas=d, noisemaps=nm, cosmics_masks=cosm, seeing_pixels=3.5, id=f))
    batched = prepare_psf_stamps_batched(frames)
    for fr, (bd, bn, bm, bk) in zip(frames, batched):
        hd, hn, hm, hk = prepare_psf_stamps(fr['datas'], fr['noisemaps'], fr['cosmics_masks'])
        np.testing.assert_array_equal(bk, hk)
        np.testing.assert_array_equal(bm, hm)
        np.testing.assert_allclose(bd, hd, rtol=0, atol=0)
        np.testing.assert_allclose(bn, hn, rtol=1e-7)
    assert batched[1][3].tolist() == [True, True, False, True] and batched[2][3].sum() == 0
    out = model_psfs_of_frames(frames, psf_n_iter_analytic=20, psf_n_iter_pixels=30)
    assert out[2][1] is None
    for f in (0, 1):
        res = out[f][1]
        assert res['narrow_psf'].shape == (64, 64) and np.isfinite(res['chi2'])
        assert len(res['adabelief_extra_fields']['loss_history']) == 30
        assert res['stars_kept'].sum() == (4 if f == 0 else 3)
