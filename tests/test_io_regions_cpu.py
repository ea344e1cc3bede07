"""The on-disk input contracts on a dict-of-arrays stand-in that uses the reference's dataset names
(cutout_making.py:156-266, psf_modelling.py:113-127,190-202, roi_file_preparation.py:215-229): packing into dense padded
batches, ragged star lists, mask polarity, coordinate rescaling, PSF bookkeeping, the per-star gather and the ROI file.
h5py is not installed in the build container: the same code reads an open h5py file through the same four operations."""
import numpy as np
import pytest

from lightcurver_amd.io import regions as R


def _fake_regions(F=4, n=8, seed=0):
    rng = np.random.default_rng(seed)
    root, sel = {}, []
    for f in range(F):
        rel = f'night{f // 2}/img_{f:03d}.fits'
        ids = [str(1000 + s) for s in range(3 + (f % 2))]          # ragged: 3 or 4 stars
        g = dict(frame_shape=np.array([200, 300]), data={}, noisemap={}, cosmicsmask={}, wcs={}, image_pixel_coordinates={})
        for name in ids + ['ROI']:
            g['data'][name] = rng.normal(size=(n, n)).astype(np.float32)
            g['noisemap'][name] = rng.uniform(0.5, 1.5, size=(n, n)).astype(np.float32)
            m = np.zeros((n, n), bool)
            m[rng.integers(n), rng.integers(n)] = True
            g['cosmicsmask'][name] = m
            g['wcs'][name] = 'WCS'
            g['image_pixel_coordinates'][name] = np.array([rng.uniform(0, 299), rng.uniform(0, 199)])
        root[rel] = g
        sel.append((rel, ids))
    return root, sel


def test_rescale_image_coordinates_known_values():
    # lightcurver/utilities/image_coordinates.py: corners map to -+0.5 * (1 - 1/dim), the centre to 0
    xy = np.array([[0.0, 0.0], [299.0, 199.0], [149.5, 99.5]])
    out = R.rescale_image_coordinates(xy, (200, 300))
    assert np.allclose(out, [[-149.5 / 300, -99.5 / 200], [149.5 / 300, 99.5 / 200], [0.0, 0.0]])


def test_read_psf_batch_packs_ragged_frames():
    root, sel = _fake_regions()
    b = R.read_psf_batch(root, sel)
    F, S, n = 4, 4, 8
    assert b['data'].shape == b['noisemap'].shape == b['cosmics'].shape == (F, S, n, n)
    assert b['data'].dtype == np.float32 and b['cosmics'].dtype == bool
    assert list(b['n_stars']) == [3, 4, 3, 4]
    for f, (rel, ids) in enumerate(sel):
        for s, name in enumerate(ids):
            assert np.array_equal(b['data'][f, s], root[rel]['data'][name])
            assert np.array_equal(b['cosmics'][f, s], root[rel]['cosmicsmask'][name])        # True = flagged, as on disk
            assert np.allclose(b['positions'][f, s], R.rescale_image_coordinates(root[rel]['image_pixel_coordinates'][name], (200, 300)))
        assert np.all(np.isnan(b['data'][f, len(ids):])) and np.all(b['cosmics'][f, len(ids):])  # padding is masked out
    frames = R.frames_for_psf_model(b, seeing_pixels=np.full(F, 3.2))
    assert [fr['datas'].shape[0] for fr in frames] == [3, 4, 3, 4] and frames[1]['seeing_pixels'] == 3.2
    assert frames[0]['stamp_coordinates'].shape == (3, 2)
    with pytest.raises(ValueError):
        R.read_psf_batch(root, [])


def test_psf_bookkeeping_and_star_gather():
    root, sel = _fake_regions()
    ref = R.psf_reference_name(['1002', '1000', '1001'])
    assert ref == 'psf_100010011002'
    N = 16
    for rel, _ in sel:
        res = dict(narrow_psf=np.full((N, N), 1.0 / N ** 2), full_psf=np.full((N, N), 1.0 / N ** 2),
                   kwargs_psf=dict(kwargs_distortion={'dilation_x': np.array([0.01, 0.0, 0.0])}))
        R.write_psf_result(root, rel, ref, res, 2)
        R.write_psf_result(root, rel, ref, res, 2)            # replacing an existing group, as the reference does
        assert set(root[rel][ref]) == {'narrow_psf', 'full_psf', 'subsampling_factor', 'distortion'}
        assert list(root[rel][ref]['subsampling_factor']) == [2]
    rels = [rel for rel, _ in sel]
    data, noise, mask, psf = R.read_star_epochs(root, rels, 1001, [ref] * 4)
    assert data.shape == noise.shape == mask.shape == (4, 8, 8) and psf.shape == (4, N, N) and mask.dtype == bool
    seen = []
    fake = lambda narrow_psf, kwargs_distortion, star_xy_coordinates: seen.append((sorted(kwargs_distortion), star_xy_coordinates)) or narrow_psf
    R.read_star_epochs(root, rels, 1001, [ref] * 4, field_distortion=True, apply_distortion=fake)
    assert len(seen) == 4 and seen[0][0] == ['dilation_x'] and np.all(np.abs(seen[0][1]) <= 0.5)


def test_read_roi_file():
    E, n, ss = 5, 8, 2
    rng = np.random.default_rng(1)
    f = dict(frame_id=np.arange(E), data=rng.normal(size=(E, n, n)), noisemap=np.ones((E, n, n)), psf=np.ones((E, n * ss, n * ss)),
             seeing=np.ones(E), sky_level_electron_per_second=np.zeros(E), mjd=np.arange(E) + 6e4,
             global_zeropoint=np.array(25.0), global_zeropoint_scatter=np.array(0.01),
             relative_normalization_error=np.full(E, 0.01), wcs=np.array(['w'] * E), pixel_scale=np.full(E, 0.2),
             subsampling_factor=np.full(E, ss), angle_to_north=np.zeros(E))
    out = R.read_roi_file(f)
    assert out['subsampling'] == ss and set(R.ROI_FILE_KEYS) <= set(out)
    bad = dict(f, subsampling_factor=np.array([2, 2, 3, 2, 2]))
    with pytest.raises(ValueError):
        R.read_roi_file(bad)
    with pytest.raises(KeyError):
        R.read_roi_file({k: v for k, v in f.items() if k != 'psf'})
    with pytest.raises(ImportError):
        R.read_roi_file('/nonexistent/cutouts.h5')   # a path needs h5py, which this container lacks
