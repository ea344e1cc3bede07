"""Array-level restatement of the per-frame work of the reference's ``model_all_psfs``
(lightcurver/processes/psf_modelling.py:126-160 stamp preparation, :164-171 PSF build, :177-180 and
:205-208 quality numbers) with the whole list of frames fitted in one device batch.  Reading the stamps
from regions.h5 is in ``lightcurver_amd/io/regions.py``, the source masking of :35-61 in ``source_masking.py``; plots
and the sqlite bookkeeping stay with the caller (out of scope)."""
import numpy as np

from ..starred.procedures.psf_routines import build_psf_batch


def mask_surrounding_stars(data, noisemap, thresh=3.0, minarea=15, deblend_cont=0.001):
    """Mask (False) every detected object of a stamp except the one closest to the stamp centre
    (psf_modelling.py:35-61).  The reference segments the stamp with ``sep.extract(data, thresh=3, err=noisemap,
    minarea=15, segmentation_map=True, deblend_cont=0.001)``; ``sep`` is not a dependency here, the detection /
    multi-threshold de-blending / segmentation are restated in ``processes/source_masking.py``."""
    from .source_masking import extract
    objects, seg_map = extract(data, noisemap, thresh=thresh, minarea=minarea, deblend_cont=deblend_cont)
    mask = np.ones(seg_map.shape, dtype=bool)
    if len(objects) == 0:
        return mask
    center_y = (seg_map.shape[0] - 1) / 2.0
    center_x = (seg_map.shape[1] - 1) / 2.0
    distances = np.hypot(objects['x'] - center_x, objects['y'] - center_y)
    central = int(np.argmin(distances))
    for i in range(len(objects)):
        if i != central:
            mask[seg_map == i + 1] = False
    return mask


def prepare_psf_stamps(datas, noisemaps, cosmics_masks, automatic_masks=None, mask_threshold_fraction=0.4):
    """Masks and clean-up of one frame's star stamps.

    cosmics_masks: True where a cosmic / bad pixel was flagged (as stored in regions.h5);
    automatic_masks: True for good pixels (neighbouring objects masked out), optional.
    Returns (datas, noisemaps, masks, keep): arrays restricted to the stamps that survive the
    more-than-40 %-masked cut (psf_modelling.py:144-153), masks True = usable pixel, keep = boolean selector.
    """
    datas = np.array(datas, dtype=np.float64)
    noisemaps = np.array(noisemaps, dtype=np.float64)
    good = ~np.asarray(cosmics_masks, dtype=bool)
    if automatic_masks is not None:
        good = good & np.asarray(automatic_masks, dtype=bool)
    both_nan = np.isnan(datas) & np.isnan(noisemaps)
    datas[both_nan] = 0.0
    noisemaps[both_nan] = 1.0
    good[both_nan] = False
    n_masked = np.sum(~good, axis=(1, 2))
    keep = ~(n_masked > mask_threshold_fraction * datas.shape[1] * datas.shape[2])
    return datas[keep], noisemaps[keep], good[keep], keep


def prepare_psf_stamps_batched(frames, mask_threshold_fraction=0.4):
    """prepare_psf_stamps for every frame at once: the stamps of all frames go through one launch of the fused
    device pass (lc_prepare_stamps: NaN clean-up, masks, masked-pixel counts), then the per-frame 40 % cut is
    applied to the counts.  Same return convention as prepare_psf_stamps, one tuple per frame."""
    from .preprocessing import prepare_stamps
    sizes = [len(fr['datas']) for fr in frames]
    if sum(sizes) == 0:
        return [(np.zeros((0, 0, 0)), np.zeros((0, 0, 0)), np.zeros((0, 0, 0), bool), np.zeros(0, bool)) for _ in frames]
    datas = np.concatenate([np.asarray(fr['datas'], dtype=np.float32) for fr in frames if len(fr['datas'])])
    noise = np.concatenate([np.asarray(fr['noisemaps'], dtype=np.float32) for fr in frames if len(fr['datas'])])
    bad = np.concatenate([_flagged(fr) for fr in frames if len(fr['datas'])])
    out = prepare_stamps(datas, noisemap=noise, bad=bad, nan_noise=1.0, want=('data', 'noisemap', 'weight'))
    npix = datas.shape[1] * datas.shape[2]
    keep_all = ~(out['masked_count'] > mask_threshold_fraction * npix)
    # usable pixel = not flagged and not NaN-in-both (psf_modelling.py:135-143); a pixel that is NaN in only one
    # input is left to the fit's own non-finite handling, as in the reference
    both_nan = np.isnan(datas) & np.isnan(noise)
    good_all = ~bad & ~both_nan
    res, o = [], 0
    for sz in sizes:
        sl = slice(o, o + sz)
        keep = keep_all[sl]
        res.append((out['data'][sl][keep].astype(np.float64), out['noisemap'][sl][keep].astype(np.float64),
                    good_all[sl][keep], keep))
        o += sz
    return res


def _flagged(fr):
    bad = np.asarray(fr['cosmics_masks'], dtype=bool)
    if fr.get('automatic_masks') is not None:
        bad = bad | ~np.asarray(fr['automatic_masks'], dtype=bool)
    return bad


def relative_loss_differential(loss_history):
    """(max - min of the last 10 %) / (max - min of the first 90 %) of the loss curve (:205-208)."""
    lh = np.asarray(loss_history, dtype=np.float64)
    cut = int(0.9 * lh.size)
    start = np.nanmax(lh[:cut]) - np.nanmin(lh[:cut])
    end = np.nanmax(lh[cut:]) - np.nanmin(lh[cut:])
    return float(end / start)


def model_psfs_of_frames(frames, subsampling_factor=2, psf_n_iter_analytic=100, psf_n_iter_pixels=3000,
                         field_distortion=False, **build_kwargs):
    """frames: iterable of dicts with 'datas', 'noisemaps', 'cosmics_masks' (and optionally
    'automatic_masks', 'seeing_pixels', 'pixel_scale', 'id').  Frames whose stamps are all rejected are
    skipped (psf_modelling.py:154-160).  Returns a list of (frame, result-or-None) with the quantities the
    reference stores: narrow_psf, full_psf, chi2, relative_loss_differential, fwhm_moffat_pixels."""
    prepared, index = [], []
    cleaned = prepare_psf_stamps_batched(frames)
    for k, fr in enumerate(frames):
        d, nmap, m, keep = cleaned[k]
        if len(d) == 0:
            continue
        prepared.append((d, nmap, m, float(fr.get('seeing_pixels', 3.0)), keep))
        index.append(k)
    out = [(fr, None) for fr in frames]
    if not prepared:
        return out
    results = build_psf_batch([p[0] for p in prepared], [p[1] for p in prepared], subsampling_factor,
                              masks=[p[2] for p in prepared], n_iter_analytic=psf_n_iter_analytic,
                              n_iter_adabelief=psf_n_iter_pixels, guess_method_star_position='center',
                              guess_fwhm_pixels=np.array([p[3] for p in prepared]), field_distortion=field_distortion,
                              **build_kwargs)
    for k, p, res in zip(index, prepared, results):
        km = res['kwargs_psf']['kwargs_moffat']
        res['fwhm_moffat_pixels'] = float((0.5 * (km['fwhm_x'] + km['fwhm_y'])).item())
        res['relative_loss_differential'] = relative_loss_differential(res['adabelief_extra_fields']['loss_history'])
        res['stars_kept'] = p[4]
        out[k] = (frames[k], res)
    return out
