"""Golden tables for the light-curve post-processing, captured from the
REFERENCE module itself, which is importable in the build container (pure numpy / pandas / scipy):
    python tests/golden/make_postprocessing_golden.py
Only inputs and outputs are stored (CSV); the reference source never travels."""
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, '/root/reference')
from lightcurver.utilities.lightcurves_postprocessing import convert_flux_to_magnitude, group_observations  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(99)
nights = np.repeat(np.arange(59000, 59012, 1.0), rng.integers(1, 6, 12))
mjd = nights + rng.uniform(0.30, 0.45, nights.size)
n = mjd.size
df = pd.DataFrame({'mjd': mjd, 'zeropoint': 25.2, 'seeing': rng.uniform(0.7, 1.5, n), 'reduced_chi2': rng.uniform(0.8, 1.3, n)})
for ps, base in (('A', 200.0), ('B', 40.0), ('C', 3.0)):
    err = rng.uniform(0.5, 2.0, n) * np.sqrt(base) / 3
    df[f'{ps}_flux'] = base * (1 + 0.05 * np.sin(mjd / 3.0)) + err * rng.standard_normal(n)
    df[f'{ps}_d_flux'] = err
df.loc[5, 'A_flux'] *= 3.0   # an outlier for the sigma clip
df.loc[9, 'C_flux'] = -1.0   # negative flux -> NaN magnitude
df = df.sample(frac=1.0, random_state=1).reset_index(drop=True)  # unsorted input
df.to_csv(os.path.join(HERE, 'postproc_input.csv'), index=False)
grouped = group_observations(df)
grouped.to_csv(os.path.join(HERE, 'postproc_grouped.csv'), index=False)
convert_flux_to_magnitude(grouped).to_csv(os.path.join(HERE, 'postproc_mags.csv'), index=False)
convert_flux_to_magnitude(df).to_csv(os.path.join(HERE, 'postproc_mags_per_epoch.csv'), index=False)
print('written', len(df), 'epochs,', len(grouped), 'nights')
