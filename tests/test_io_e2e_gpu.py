"""SURVEY.md 8(f) row f2 end to end on the device: regions layout -> ``read_psf_batch`` -> ``model_psfs_of_frames``
(= ``build_psf_batch``) -> ``write_psf_result`` -> ``read_star_epochs`` -> star photometry, against the same arrays handed
to the same functions directly: bit for bit.  The regions node is the dict-of-arrays stand-in with the reference's dataset
names (cutout_making.py:156-266, psf_modelling.py:113-127,190-202); with h5py present (it is not in the build container)
the same chain runs through a real file."""
import numpy as np
import pytest

from lightcurver_amd.io import regions as R
from lightcurver_amd.synthetic import make_psf_dataset

pytestmark = pytest.mark.gpu

F, S, n, ss = 5, 4, 16, 2
FRAME_SHAPE = (400, 600)


def _regions_from_dataset(ds, h5=None):
    """The synthetic stamps in the layout cutout_making.py writes; ragged star lists (frame 1 has one star less)."""
    rng = np.random.default_rng(5)
    root = h5 if h5 is not None else {}
    sel = []
    for f in range(F):
        rel = f'night{f // 2}/img_{f:03d}.fits'
        ids = [str(4000 + s) for s in range(S - (1 if f == 1 else 0))]
        if h5 is not None:
            g = root.create_group(rel)
            sub = {k: g.create_group(k) for k in ('data', 'noisemap', 'cosmicsmask', 'image_pixel_coordinates')}
            g['frame_shape'] = np.array(FRAME_SHAPE)
        else:
            g = root.setdefault(rel, dict(frame_shape=np.array(FRAME_SHAPE)))
            sub = {k: g.setdefault(k, {}) for k in ('data', 'noisemap', 'cosmicsmask', 'image_pixel_coordinates')}
        for s, name in enumerate(ids):
            sub['data'][name] = ds['data'][f, s]
            sub['noisemap'][name] = ds['noisemap'][f, s]
            sub['cosmicsmask'][name] = ~ds['masks'][f, s]                      # on disk: True = flagged
            sub['image_pixel_coordinates'][name] = np.array([rng.uniform(20, 580), rng.uniform(20, 380)])
        sel.append((rel, ids))
    return root, sel


def _chain(root, sel, ds, n_pix_iter=60):
    from lightcurver_amd.processes.psf_modelling import model_psfs_of_frames
    batch = R.read_psf_batch(root, sel)
    frames = R.frames_for_psf_model(batch, seeing_pixels=ds['fwhm_guess'])
    fitted = model_psfs_of_frames(frames, subsampling_factor=ss, psf_n_iter_analytic=20, psf_n_iter_pixels=n_pix_iter)
    refs = []
    for (rel, ids), (fr, res) in zip(sel, fitted):
        assert res is not None
        ref = R.psf_reference_name(ids)
        R.write_psf_result(root, rel, ref, res, ss)
        refs.append(ref)
    return batch, fitted, refs


def test_reader_to_fit_to_writer_to_star_gather_equals_the_direct_calls():
    from lightcurver_amd.processes.psf_modelling import model_psfs_of_frames
    from lightcurver_amd.processes.star_photometry import do_one_star_forward_modelling, prepare_star_epochs
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=77)
    root, sel = _regions_from_dataset(ds)
    batch, fitted, refs = _chain(root, sel, ds)
    assert batch['data'].shape == (F, S, n, n) and list(batch['n_stars']) == [4, 3, 4, 4, 4]

    # the direct call: the same stamps as arrays, never through the regions layout
    direct_frames = [dict(datas=ds['data'][f, :len(ids)].astype(np.float64), noisemaps=ds['noisemap'][f, :len(ids)].astype(np.float64),
                          cosmics_masks=~ds['masks'][f, :len(ids)], seeing_pixels=float(ds['fwhm_guess'][f]))
                     for f, (rel, ids) in enumerate(sel)]
    direct = model_psfs_of_frames(direct_frames, subsampling_factor=ss, psf_n_iter_analytic=20, psf_n_iter_pixels=60)
    for (fr, res), (_, dres), (rel, ids), ref in zip(fitted, direct, sel, refs):
        for key in ('narrow_psf', 'full_psf', 'residuals'):
            assert np.array_equal(np.asarray(res[key]), np.asarray(dres[key])), key
        assert res['chi2'] == dres['chi2']
        assert np.array_equal(res['adabelief_extra_fields']['loss_history'], dres['adabelief_extra_fields']['loss_history'])
        # what the writer stored is what the fit returned (psf_modelling.py:190-202)
        g = root[rel][ref]
        assert np.array_equal(np.asarray(g['narrow_psf']), np.asarray(res['narrow_psf']))
        assert list(np.asarray(g['subsampling_factor'])) == [ss]

    # per-star gather over the frames (star_photometry.py:276-306), then the joint fit of that star
    star = '4001'
    rels = [rel for rel, _ in sel]
    data, noise, mask, psf = R.read_star_epochs(root, rels, star, refs)
    assert data.shape == (F, n, n) and psf.shape == (F, n * ss, n * ss) and mask.dtype == bool
    for f in range(F):
        assert np.array_equal(data[f], ds['data'][f, 1]) and np.array_equal(mask[f], ~ds['masks'][f, 1])
        assert np.array_equal(psf[f], np.asarray(fitted[f][1]['narrow_psf']))
    d1, n1 = prepare_star_epochs(data, noise, mask)
    out = do_one_star_forward_modelling(d1, n1, psf, ss, n_iter=100, starlet_global_background=False)
    d2, n2 = prepare_star_epochs(ds['data'][:, 1], ds['noisemap'][:, 1], ~ds['masks'][:, 1])
    psf_direct = np.array([np.asarray(dres['narrow_psf']) for _, dres in direct])
    ref_out = do_one_star_forward_modelling(d2, n2, psf_direct, ss, n_iter=100, starlet_global_background=False)
    assert np.array_equal(out['fluxes'], ref_out['fluxes']) and out['chi2'] == ref_out['chi2']
    assert np.all(np.isfinite(out['fluxes'])) and out['fluxes'].shape == (F,)
    assert np.array_equal(out['fluxes_uncertainties'], ref_out['fluxes_uncertainties'])


def test_the_same_chain_through_a_real_hdf5_file(tmp_path):
    h5py = pytest.importorskip('h5py')
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=77)
    path = tmp_path / 'regions.h5'
    with h5py.File(path, 'w') as f:
        _, sel = _regions_from_dataset(ds, h5=f)
    with h5py.File(path, 'r+') as f:
        batch_h5, fitted_h5, refs = _chain(f, sel, ds)
    root, sel2 = _regions_from_dataset(ds)
    batch, fitted, _ = _chain(root, sel2, ds)
    assert np.array_equal(batch_h5['data'], batch['data'], equal_nan=True)
    for (_, a), (_, b) in zip(fitted_h5, fitted):
        assert np.array_equal(np.asarray(a['narrow_psf']), np.asarray(b['narrow_psf']))
    rels = [rel for rel, _ in sel]
    from_path = R.read_star_epochs(str(path), rels, '4001', refs)      # a path: the reader opens the file itself
    from_dict = R.read_star_epochs(root, rels, '4001', refs)
    for a, b in zip(from_path, from_dict):
        assert np.array_equal(a, b)
