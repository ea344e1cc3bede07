"""Light-curve post-processing: the reference's own numeric tests restated
(tests/test_products_handling/test_grouping.py:7-59, test_magnitude_errors.py:9-81) and golden tables captured
from the reference module itself (tests/golden/make_postprocessing_golden.py): this row IS pinned."""
import os

import numpy as np
import pandas as pd
import pytest

from lightcurver_amd.utilities.lightcurves_postprocessing import convert_flux_to_magnitude, group_observations

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_grouping_multiple_observations():
    df = pd.DataFrame({'mjd': [1.0, 1.2, 2.5, 2.6], 'A_flux': [10.0, 12.0, 20.0, 22.0], 'A_d_flux': [1.0, 1.0, 2.0, 2.0],
                       'other': [100, 200, 300, 400]})
    result = group_observations(df, threshold=0.8)
    assert len(result) == 2
    np.testing.assert_almost_equal(result.loc[0, 'A_flux'], 11.0, decimal=3)
    np.testing.assert_almost_equal(result.loc[1, 'A_flux'], 21.0, decimal=3)
    np.testing.assert_almost_equal(result.loc[0, 'other'], 150.0, decimal=3)
    np.testing.assert_almost_equal(result.loc[1, 'other'], 350.0, decimal=3)


def test_single_observation_group_and_last_group():
    result = group_observations(pd.DataFrame({'mjd': [1.0], 'A_flux': [10.0], 'A_d_flux': [1.0]}), threshold=0.8)
    assert len(result) == 1
    np.testing.assert_almost_equal(result.loc[0, 'A_flux'], 10.0, decimal=3)
    np.testing.assert_almost_equal(result.loc[0, 'A_d_flux'], 1.0, decimal=3)
    assert result.loc[0, 'A_count_flux'] == 1
    result = group_observations(pd.DataFrame({'mjd': [1.0, 1.2, 3.0], 'A_flux': [10.0, 12.0, 20.0],
                                              'A_d_flux': [1.0, 1.0, 2.0]}), threshold=0.8)
    assert len(result) == 2
    np.testing.assert_almost_equal(result.loc[0, 'A_flux'], 11.0, decimal=5)
    np.testing.assert_almost_equal(result.loc[1, 'A_flux'], 20.0, decimal=3)
    np.testing.assert_almost_equal(result.loc[1, 'mjd'], 3.0, decimal=5)


def test_convert_flux_to_magnitude_reference_values():
    df = pd.DataFrame({'A_flux': [100, 50, 10, 5], 'A_d_flux': [10, 5, 2, 6], 'A_scatter_flux': [8, 4, 1.5, 3],
                       'zeropoint': [25, 25, 25, 25]})
    expected = {'A_mag': [20.0, 20.7526, 22.5, 23.253], 'A_d_mag_down': [0.1035, 0.1035, 0.1980, 0.856],
                'A_d_mag_up': [0.1144, 0.1142, 0.2423, np.nan], 'A_scatter_mag_down': [0.0835, 0.0835, 0.152, 0.510],
                'A_scatter_mag_up': [0.090, 0.090, 0.176, 0.995]}
    res = convert_flux_to_magnitude(df)
    for col, vals in expected.items():
        for i, v in enumerate(vals):
            if np.isnan(v):
                assert np.isnan(res.at[i, col]), (col, i)
            else:
                assert abs(res.at[i, col] - v) < 1e-2, (col, i)


def test_missing_zeropoint_warns():
    with pytest.warns(RuntimeWarning):
        out = convert_flux_to_magnitude(pd.DataFrame({'A_flux': [10.0], 'A_d_flux': [1.0]}))
    assert out.loc[0, 'zeropoint'] == 0.0 and abs(out.loc[0, 'A_mag'] + 2.5) < 1e-12


def _same(a, b):
    assert list(a.columns) == list(b.columns)
    for c in a.columns:
        np.testing.assert_allclose(a[c].to_numpy(dtype=float), b[c].to_numpy(dtype=float), rtol=1e-10, atol=1e-12,
                                   equal_nan=True, err_msg=c)


def test_golden_tables_from_the_reference_module():
    df = pd.read_csv(os.path.join(GOLD, 'postproc_input.csv'))
    grouped = group_observations(df)
    _same(grouped, pd.read_csv(os.path.join(GOLD, 'postproc_grouped.csv')))
    _same(convert_flux_to_magnitude(grouped), pd.read_csv(os.path.join(GOLD, 'postproc_mags.csv')))
    _same(convert_flux_to_magnitude(df), pd.read_csv(os.path.join(GOLD, 'postproc_mags_per_epoch.csv')))
