"""On-disk input contracts of the hot path, restated as batched readers.

The reference keeps every cutout in one HDF5 file, ``regions.h5``, written by
lightcurver/processes/cutout_making.py:156-266 and read back per frame inside serial Python loops
(psf_modelling.py:92,113-127; star_photometry.py:276-304; roi_file_preparation.py:157-185):

    /{image_relpath}/frame_shape                                  (2,)  shape of the full frame (rows, columns)
    /{image_relpath}/data/{gaia_id | 'ROI'}                       (n, n) float
    /{image_relpath}/noisemap/{gaia_id | 'ROI'}                   (n, n) float
    /{image_relpath}/cosmicsmask/{gaia_id | 'ROI'}                (n, n) bool, True = flagged pixel
    /{image_relpath}/wcs/{gaia_id | 'ROI'}                        str
    /{image_relpath}/image_pixel_coordinates/{gaia_id | 'ROI'}    (2,)  (x, y) of the cutout centre in the frame
    /{image_relpath}/{psf_ref}/narrow_psf, full_psf               (N, N)           written by psf_modelling.py:190-202
    /{image_relpath}/{psf_ref}/subsampling_factor                 (1,)
    /{image_relpath}/{psf_ref}/distortion/{key}                   kwargs_distortion of build_psf

and the calibrated ROI stack in ``cutouts_{hash}_{roi}.h5`` (roi_file_preparation.py:215-229): frame_id, data, noisemap,
psf, seeing, sky_level_electron_per_second, mjd, global_zeropoint, global_zeropoint_scatter,
relative_normalization_error, wcs, pixel_scale, subsampling_factor, angle_to_north.

Once the fit of a whole dataset takes a fraction of a second on the device, that per-frame loop is
the cost; the functions below read every frame of a selection in one pass into the dense, padded ``[F][S][n][n]``
buffers ``build_psf_batch`` / ``lc_psf_batch_create`` take.  ``regions`` may be an open ``h5py.File`` / ``h5py.Group``, a
path (h5py is imported only then; it is not installed in the build container) or any nested mapping with the same keys
(what the tests use): only ``node[key]``, ``key in node``, ``node.keys()`` and ``np.asarray(leaf)`` are relied on.
"""
import contextlib
import os

import numpy as np


def rescale_image_coordinates(xy_coordinates_array, image_shape):
    """lightcurver/utilities/image_coordinates.py:6-27: origin to the frame centre, unit = frame size, so that the
    coordinates span [-0.5, 0.5] (x along columns, y along rows)."""
    dims = np.asarray(image_shape, dtype=np.float64)[::-1]
    return (np.asarray(xy_coordinates_array, dtype=np.float64) - (dims - 1.0) / 2.0) / dims


@contextlib.contextmanager
def open_regions(regions, mode='r'):
    """Yields a node to read from: the mapping / h5py object itself, or the file behind a path."""
    if isinstance(regions, (str, os.PathLike)):
        try:
            import h5py
        except ImportError as e:  # the layout above is restated, the HDF5 library itself is the caller's
            raise ImportError('reading a regions file from a path needs h5py; pass an open file or a mapping instead') from e
        with h5py.File(regions, mode) as f:
            yield f
    else:
        yield regions


def _leaf(node):
    a = node[...] if hasattr(node, 'shape') and not isinstance(node, np.ndarray) else node
    return np.asarray(a)


def read_psf_batch(regions, frames, dtype=np.float32):
    """One pass over the selection.  frames: iterable of (image_relpath, [gaia ids of the PSF stars of that frame])
    - what select_stars_for_a_frame yields per frame in the reference (psf_modelling.py:94-104).

    Returns a dict of dense arrays, stars padded to S = the largest star count (``n_stars[f]`` are valid):
      data, noisemap  (F, S, n, n) ``dtype`` (padding = NaN, which build_psf_batch / lc_psf_batch_create mask out)
      cosmics         (F, S, n, n) bool, True = flagged (padding = True)
      positions       (F, S, 2) rescaled frame coordinates of the stamps (field distortion), frame_shape (F, 2)
      n_stars (F,), names: list of lists, image_relpath: list.
    """
    frames = [(str(rel), [str(g) for g in ids]) for rel, ids in frames]
    F = len(frames)
    S = max((len(ids) for _, ids in frames), default=0)
    if F == 0 or S == 0:
        raise ValueError('nothing to read: no frame, or no star in any frame')
    with open_regions(regions) as root:
        first_rel, first_ids = next((rel, ids) for rel, ids in frames if ids)
        n = _leaf(root[first_rel]['data'][first_ids[0]]).shape[-1]
        data = np.full((F, S, n, n), np.nan, dtype)
        noise = np.full((F, S, n, n), np.nan, dtype)
        cosmics = np.ones((F, S, n, n), bool)
        pos = np.zeros((F, S, 2), np.float64)
        shapes = np.zeros((F, 2), np.int64)
        for f, (rel, ids) in enumerate(frames):
            g = root[rel]
            shapes[f] = _leaf(g['frame_shape'])
            dg, ng, mg, pg = g['data'], g['noisemap'], g['cosmicsmask'], g['image_pixel_coordinates']
            for s, name in enumerate(ids):
                d = _leaf(dg[name])
                if d.shape != (n, n):
                    raise ValueError(f'{rel}/data/{name}: stamp of shape {d.shape}, expected {(n, n)}')
                data[f, s] = d
                noise[f, s] = _leaf(ng[name])
                cosmics[f, s] = _leaf(mg[name]).astype(bool)
                pos[f, s] = _leaf(pg[name])
            if ids:
                pos[f, :len(ids)] = rescale_image_coordinates(pos[f, :len(ids)], shapes[f])
    return dict(data=data, noisemap=noise, cosmics=cosmics, positions=pos, frame_shape=shapes,
                n_stars=np.array([len(ids) for _, ids in frames]), names=[ids for _, ids in frames],
                image_relpath=[rel for rel, _ in frames])


def frames_for_psf_model(batch, seeing_pixels=None, ids=None):
    """The dense batch as the list of per-frame dicts ``processes.psf_modelling.model_psfs_of_frames`` consumes
    (datas, noisemaps, cosmics_masks with True = flagged, stamp_coordinates, seeing_pixels, id)."""
    out = []
    for f, k in enumerate(batch['n_stars']):
        k = int(k)
        fr = dict(datas=batch['data'][f, :k].astype(np.float64), noisemaps=batch['noisemap'][f, :k].astype(np.float64),
                  cosmics_masks=batch['cosmics'][f, :k], stamp_coordinates=batch['positions'][f, :k],
                  names=batch['names'][f], image_relpath=batch['image_relpath'][f], id=f if ids is None else ids[f])
        if seeing_pixels is not None:
            fr['seeing_pixels'] = float(np.asarray(seeing_pixels)[f])
        out.append(fr)
    return out


def psf_reference_name(star_names):
    """'psf_' + the sorted names joined (psf_modelling.py:87, star_photometry.py:287)."""
    return 'psf_' + ''.join(sorted(str(s) for s in star_names))


def write_psf_result(regions, image_relpath, psf_ref, result, subsampling_factor):
    """Bookkeeping of psf_modelling.py:189-202 on an open (writable) regions node: replaces the group of ``psf_ref``."""
    with open_regions(regions, 'r+') as root:
        g = root[image_relpath]
        if psf_ref in g.keys():
            del g[psf_ref]
        grp = g.create_group(psf_ref) if hasattr(g, 'create_group') else g.setdefault(psf_ref, {})
        grp['narrow_psf'] = np.array(result['narrow_psf'])
        grp['full_psf'] = np.array(result['full_psf'])
        grp['subsampling_factor'] = np.array([subsampling_factor])
        dist = grp.create_group('distortion') if hasattr(grp, 'create_group') else grp.setdefault('distortion', {})
        for key, value in result['kwargs_psf']['kwargs_distortion'].items():
            dist[key] = value


def read_star_epochs(regions, frames, gaia_id, psf_refs, field_distortion=False, apply_distortion=None):
    """The per-star gather of star_photometry.py:276-306 in one pass: for every (image_relpath) of ``frames`` the stamp,
    noise map and cosmics mask of ``gaia_id`` and the narrow PSF of that frame (``psf_refs[f]``), optionally resampled
    at the star's position with the frame's fitted distortion.  Returns data, noisemap (E, n, n), mask (E, n, n) bool
    True = flagged, psf (E, N, N)."""
    gaia_id = str(gaia_id)
    data, noise, mask, psf = [], [], [], []
    with open_regions(regions) as root:
        for rel, ref in zip(frames, psf_refs):
            g = root[str(rel)]
            data.append(_leaf(g['data'][gaia_id]))
            noise.append(_leaf(g['noisemap'][gaia_id]))
            mask.append(_leaf(g['cosmicsmask'][gaia_id]).astype(bool))
            narrow = _leaf(g[ref]['narrow_psf'])
            if field_distortion:
                if apply_distortion is None:
                    from ..starred.psf.psf import apply_distortion
                kw = {k: _leaf(g[ref]['distortion'][k]) for k in g[ref]['distortion'].keys()}
                position = rescale_image_coordinates(_leaf(g['image_pixel_coordinates'][gaia_id]), _leaf(g['frame_shape']))
                narrow = apply_distortion(narrow_psf=narrow, kwargs_distortion=kw, star_xy_coordinates=position)
            psf.append(narrow)
    return np.array(data), np.array(noise), np.array(mask), np.array(psf)


ROI_FILE_KEYS = ('frame_id', 'data', 'noisemap', 'psf', 'seeing', 'sky_level_electron_per_second', 'mjd',
                 'global_zeropoint', 'global_zeropoint_scatter', 'relative_normalization_error', 'wcs', 'pixel_scale',
                 'subsampling_factor', 'angle_to_north')


def read_roi_file(roi_file):
    """The calibrated ROI stack written by roi_file_preparation.py:215-229 as the arguments of
    ``processes.roi_modelling.model_roi_cutouts``: everything in ROI_FILE_KEYS, plus ``subsampling`` (the single value
    the reference takes with np.unique at roi_modelling.py:166-170)."""
    with open_regions(roi_file) as root:
        missing = [k for k in ('data', 'noisemap', 'psf', 'subsampling_factor') if k not in root.keys()]
        if missing:
            raise KeyError(f'ROI file lacks the datasets {missing}')
        out = {k: _leaf(root[k]) for k in ROI_FILE_KEYS if k in root.keys()}
    ss = np.unique(out['subsampling_factor'])
    if ss.size != 1:
        raise ValueError(f'all epochs must share one subsampling factor, found {ss}')
    out['subsampling'] = int(ss[0])
    E, n, _ = out['data'].shape
    N = out['subsampling'] * n
    if out['noisemap'].shape != (E, n, n) or out['psf'].shape != (E, N, N):
        raise ValueError('data / noisemap / psf shapes are inconsistent')
    return out
