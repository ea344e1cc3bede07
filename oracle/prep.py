"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy, float64) of the stamp pre-processing the reference does
on the host before its fits.  PARITY UNPINNED: the reference modules holding these lines
import astropy / h5py / starred, none of which exist here, and none of its tests pins these numbers; the
functions below follow the cited lines literally.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import this package.
"""
import numpy as np


def noisemap_from_rms(data, rms, exptime):
    """lightcurver/processes/cutout_making.py:43-51: data in e-/s, rms in e-/s, exptime in s -> noise map in e-/s."""
    data = np.asarray(data, dtype=np.float64)
    t = np.asarray(exptime, dtype=np.float64).reshape(-1, *([1] * (data.ndim - 1)))
    r = np.asarray(rms, dtype=np.float64).reshape(t.shape)
    electrons = t * data
    with np.errstate(invalid='ignore'):
        noise = ((t * r) ** 2 + np.abs(electrons)) ** 0.5
        noise[noise < 1e-7] = 1e-7
    return noise / t


def prepare(data, noisemap=None, rms=None, exptime=None, coefficient=None, bad=None, nan_noise=1.0, noise_boost=0.0,
            boost_whole_stamp=False):
    """data, noisemap, bad: (K, ...) stacks.  Returns (data, noisemap, weight, masked_count).

    coefficient: roi_file_preparation.py:162-164; both-NaN pixels: psf_modelling.py:139-143 (nan_noise = 1) /
    roi_file_preparation.py:194-196, star_photometry.py:309-311 (nan_noise = 1e7); flagged pixels:
    roi_file_preparation.py:201 (per pixel) / star_photometry.py:316 (whole epoch, once); masked count:
    psf_modelling.py:144-149."""
    d = np.array(data, dtype=np.float64)
    K = d.shape[0]
    s = np.array(noisemap, dtype=np.float64) if noisemap is not None else noisemap_from_rms(d, rms, exptime)
    if coefficient is not None:
        c = np.asarray(coefficient, dtype=np.float64).reshape(K, *([1] * (d.ndim - 1)))
        d = d / c
        s = s / c
    both_nan = np.isnan(d) & np.isnan(s)
    d[both_nan] = 0.0
    s[both_nan] = nan_noise
    flagged = np.zeros(d.shape, dtype=bool) if bad is None else np.asarray(bad).astype(bool)
    masked = flagged | both_nan
    if noise_boost > 0:
        if boost_whole_stamp:
            idx = np.unique(np.where(flagged)[0])
            s[idx] *= noise_boost
        else:
            s[flagged] *= noise_boost
    with np.errstate(all='ignore'):
        good = ~masked & np.isfinite(d) & np.isfinite(s) & (s > 0)
        w = np.where(good, 1.0 / (s * s), 0.0)
    return d, s, w, masked.reshape(K, -1).sum(axis=1).astype(np.int32)
