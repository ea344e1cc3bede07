"""Light-curve post-processing that consumes the fluxes of the joint fit: night binning with
2-sigma clipping + inverse-variance means, and flux -> magnitude conversion with asymmetric errors.
Restates the reference's lightcurver/utilities/lightcurves_postprocessing.py:8-149:
array-level cores, thin pandas wrappers with the reference's column conventions.  Pinned by the reference's
own numeric tests (tests/test_products_handling/) and by golden tables captured from the reference module
(tests/golden/make_postprocessing_golden.py).  Host code: a few hundred rows at most, nothing for the GPU."""
import warnings

import numpy as np
import pandas as pd
from scipy.stats import sigmaclip


def night_groups(mjd_sorted, threshold=0.8):
    """Index ranges [(start, stop), ...] of consecutive observations closer than ``threshold`` days."""
    mjd_sorted = np.asarray(mjd_sorted, dtype=np.float64)
    if mjd_sorted.size == 0:
        return []
    cuts = np.flatnonzero(np.diff(mjd_sorted) > threshold) + 1
    edges = np.concatenate([[0], cuts, [mjd_sorted.size]])
    return [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:])]


def combine_fluxes(flux, d_flux):
    """Clipped inverse-variance mean of one source in one group.
    Returns (mean, error of the mean, weighted scatter, number of points kept)."""
    flux = np.asarray(flux, dtype=np.float64)
    var = np.asarray(d_flux, dtype=np.float64) ** 2
    kept, lo, hi = sigmaclip(flux, low=2, high=2)
    sel = (flux >= lo) & (flux <= hi)
    v = var[sel]
    if v.size == 0 or not np.all(v > 0):
        return float('nan'), float('inf'), float('nan'), 0
    w = 1.0 / v
    mean = np.average(kept, weights=w)
    scatter = np.sqrt(np.average((kept - mean) ** 2, weights=w))
    return float(mean), float(np.sqrt(1.0 / w.sum())), float(scatter), int(v.size)


def group_observations(df, threshold=0.8):
    """Bin a per-epoch table (columns mjd, {ps}_flux, {ps}_d_flux, anything else numeric) by night."""
    table = df.sort_values(by='mjd')
    sources = sorted({c.split('_')[0] for c in table.columns if c.endswith('_flux') and not c.endswith('_d_flux')})
    flux_cols = [f'{ps}_flux' for ps in sources] + [f'{ps}_d_flux' for ps in sources]
    other_cols = [c for c in table.columns if c != 'mjd' and c not in flux_cols]
    rows = []
    for a, b in night_groups(table['mjd'].to_numpy(), threshold):
        grp = table.iloc[a:b]
        spread = grp['mjd'].std()
        row = {'mjd': grp['mjd'].mean(), 'scatter_mjd': 0.0 if np.isnan(spread) else spread}
        for c in other_cols:
            row[c] = grp[c].mean()
        for ps in sources:
            mean, err, scatter, count = combine_fluxes(grp[f'{ps}_flux'].to_numpy(), grp[f'{ps}_d_flux'].to_numpy())
            row[f'{ps}_flux'] = mean
            row[f'{ps}_d_flux'] = err
            row[f'{ps}_scatter_flux'] = scatter
            row[f'{ps}_count_flux'] = count
        rows.append(row)
    return pd.DataFrame(rows)


def flux_to_mag(flux, err, zeropoint):
    """(mag, sigma_down, sigma_up, linearised sigma): asymmetric magnitude errors from flux +/- err, NaN on
    the side where the flux bound is not positive."""
    flux = np.asarray(flux, dtype=np.float64)
    err = np.asarray(err, dtype=np.float64)
    zp = np.broadcast_to(np.asarray(zeropoint, dtype=np.float64), flux.shape)
    with np.errstate(all='ignore'):
        mag = -2.5 * np.log10(flux) + zp
        bright = np.where(flux + err > 0, -2.5 * np.log10(np.where(flux + err > 0, flux + err, 1.0)) + zp, np.nan)
        faint = np.where(flux - err > 0, -2.5 * np.log10(np.where(flux - err > 0, flux - err, 1.0)) + zp, np.nan)
        linear = 2.5 / np.log(10) * np.abs(err / flux)
    return mag, mag - bright, faint - mag, linear


def convert_flux_to_magnitude(df):
    """Add {ps}_mag, {ps}_{d|scatter}_mag_down/_up and the linearised {ps}_{d|scatter}_mag columns."""
    out = df.copy(deep=True)
    if 'zeropoint' not in out.columns:
        warnings.warn('Zeropoint column missing. Using a zeropoint of 0.', RuntimeWarning)
        out['zeropoint_used_in_conversion'] = 0.
        out['zeropoint'] = 0.
    aux = [c for c in out.columns if '_scatter_flux' in c or '_d_flux' in c or '_count' in c]
    flux_cols = [c for c in out.columns if '_flux' in c and c not in aux]
    zp = out['zeropoint'].to_numpy()
    for kind in ('d', 'scatter'):
        for col in flux_cols:
            ps = col.split('_')[0]
            err_col = f'{ps}_{kind}_flux'
            if err_col not in out.columns:
                continue
            mag, down, up, lin = flux_to_mag(out[col].to_numpy(), out[err_col].to_numpy(), zp)
            out[f'{ps}_mag'] = mag
            out[f'{ps}_{kind}_mag_down'] = down
            out[f'{ps}_{kind}_mag_up'] = up
            out[f'{ps}_{kind}_mag'] = lin
    return out
