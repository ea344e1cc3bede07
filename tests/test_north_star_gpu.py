"""End results against the tolerance BASELINE.json's north_star states (fluxes and positions within 1e-4 relative,
residual chi2 within 1e-5): the same problem run for 300 optimiser iterations by the HIP path (fp32) and by the oracle
(fp64), at the reference's learning rates.  What holds at that level: fluxes, positions and shifts of well-constrained
sources.  What does not, and why: AdaBelief with eps = 1e-16 takes sign-like steps of size ~lr wherever a gradient
component is within rounding of zero (every pixel of the grid / background once it hovers around its optimum, the
position of a source 15 x fainter than its neighbour), so fp32 and fp64 trajectories decorrelate at the scale of the
learning rate in those directions; the chi2 of an unconverged fit inherits that at the 1e-4 ... 1e-3 level.  A single
evaluation at identical parameters agrees to 1e-6 (tests/test_psf_gpu.py, tests/test_joint_gpu.py).  (Against STARRED
itself the parity is unpinned; this is the statement the oracle allows.)"""
import numpy as np
import pytest

from oracle import model as om, optim as oo
from lightcurver_amd.synthetic import make_psf_dataset, make_roi_dataset
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_joint_fit_end_results(ctx):
    from lightcurver_amd.joint import JointFit
    E, M, n, ss, T = 6, 2, 16, 2, 300
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=2024)
    p = {k: np.array(v, dtype=np.float64) for k, v in ds['truth'].items()}
    rng = np.random.default_rng(1)
    p['a'] = p['a'] * rng.uniform(0.8, 1.2, p['a'].shape)
    p['c_x'] = p['c_x'] + rng.normal(0, 0.2, M)
    p['c_y'] = p['c_y'] + rng.normal(0, 0.2, M)
    p['h'] = np.zeros_like(p['h'])
    data, sig2, psf = om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])
    W = om.propagate_noise_deconv(sig2, psf, ss)
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, ctx)
    j.set_params(**p)
    j.set_loss(W=W.numpy(), lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h']
    j.set_free(free)
    j.run_adabelief(T, init_learning_rate=1e-3, schedule_learning_rate=True)  # star_photometry.py:117
    got = j.get_params()
    model, chi2_e = j.model()
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, W=W, lam_scales=1.0, lam_hf=1.0, lam_pos=100.0, lam_pts=0.01,
                                  lam_fu=10.0)
    po = {k: om.T(v) for k, v in p.items()}
    pf, lh, l0 = oo.adabelief(fn, po, free, 1e-3, T, schedule=True)
    mo = om.deconv_model(pf, psf, ss, n)
    chi2_o = (((data - mo) ** 2) / sig2).sum().item()
    print('joint: flux', H.rel_err(got['a'], pf['a'].numpy()), 'c_x', np.abs(got['c_x'] - pf['c_x'].numpy()), 'c_y',
          np.abs(got['c_y'] - pf['c_y'].numpy()), 'dx', np.abs(got['dx'] - pf['dx'].numpy()).max(), 'chi2',
          abs(chi2_e.sum() - chi2_o) / chi2_o, 'a', pf['a'].numpy()[:2])
    assert H.rel_err(got['a'], pf['a'].numpy()) < 1e-4                       # north-star level
    bright = int(np.argmax(pf['a'].numpy()[:M]))
    assert np.abs(got['c_x'] - pf['c_x'].numpy())[bright] < 1e-4 and np.abs(got['c_y'] - pf['c_y'].numpy())[bright] < 1e-3
    assert np.abs(got['dx'] - pf['dx'].numpy()).max() < 1e-4 and np.abs(got['dy'] - pf['dy'].numpy()).max() < 1e-3
    assert np.abs(got['c_x'] - pf['c_x'].numpy()).max() < 5e-3 and np.abs(got['c_y'] - pf['c_y'].numpy()).max() < 5e-3  # ~lr-scale wander of the faint source
    assert abs(chi2_e.sum() - chi2_o) / chi2_o < 1e-3
    assert lh[-1] < l0  # the fit went somewhere


def test_psf_fit_end_results(ctx):
    from lightcurver_amd.psf_batch import PsfBatch
    F, S, n, ss, T = 2, 5, 16, 2, 300
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=2025)
    rng = np.random.default_rng(2)
    plist = [H.psf_initial_params(ds, f, ss, rng, 0.2) for f in range(F)]
    b = PsfBatch(ds['data'], H.weights_from(ds), ss, ctx)
    b.set_moffat(H.moffat_array(plist))
    b.set_stars(H.stars_array(plist))
    b.set_grid(np.stack([p['B'].numpy() for p in plist]))
    Ws = []
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        Ws.append(om.propagate_noise_psf(plist[f], sig2, mask, ss))
    J = om.n_scales(n * ss)
    b.set_regularization(np.stack([w[:J].numpy() for w in Ws]), 1.0, 1.0)
    b.run_adabelief(T, init_learning_rate=1e-4, schedule_learning_rate=True)
    stars = b.get_stars()
    res = b.results()
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=Ws[f], lam_scales=1.0, lam_hf=1.0)
        pf, lh, l0 = oo.adabelief(fn, plist[f], ['B', 'a', 'x0', 'y0'], 1e-4, T, schedule=True)
        chi2_o = om.reduced_chi2(data, om.psf_model(pf, ss, n), sig2, mask)
        print('psf: flux', H.rel_err(stars[f][:, 0], pf['a'].numpy()), 'x0', np.abs(stars[f][:, 1] - pf['x0'].numpy()).max(),
              'y0', np.abs(stars[f][:, 2] - pf['y0'].numpy()).max(), 'chi2', abs(res['chi2'][f] - chi2_o) / chi2_o)
        assert H.rel_err(stars[f][:, 0], pf['a'].numpy()) < 1e-4                   # north-star level
        assert np.abs(stars[f][:, 1] - pf['x0'].numpy()).max() < 1e-3 and np.abs(stars[f][:, 2] - pf['y0'].numpy()).max() < 1e-3
        assert abs(res['chi2'][f] - chi2_o) / chi2_o < 5e-3


def test_star_photometry_end_results_meet_the_north_star_tolerances(ctx):
    """The reference's default star photometry (point source only, star_photometry.py:74-122: learning rate 1e-3 with
    the schedule): a smooth problem in a handful of parameters per epoch.  After 600 iterations fluxes and shifts agree
    with the oracle within 1e-4 relative and chi2 within 1e-5."""
    from lightcurver_amd.joint import JointFit
    E, M, n, ss, T = 6, 1, 16, 2, 600
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=2026, with_background=False)
    p = {k: np.array(v, dtype=np.float64) for k, v in ds['truth'].items()}
    rng = np.random.default_rng(3)
    p['a'] = p['a'] * rng.uniform(0.8, 1.2, p['a'].shape)
    p['h'] = np.zeros_like(p['h'])
    data, sig2, psf = om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, ctx)
    j.set_params(**p)
    j.set_loss()
    free = ['a', 'dx', 'dy', 'mean']
    j.set_free(free)
    j.run_adabelief(T, init_learning_rate=1e-3, schedule_learning_rate=True)
    got = j.get_params()
    model, chi2_e = j.model()
    po = {k: om.T(v) for k, v in p.items()}
    pf, lh, l0 = oo.adabelief(lambda q: om.deconv_loss(q, data, sig2, psf, ss), po, free, 1e-3, T, schedule=True)
    mo = om.deconv_model(pf, psf, ss, n)
    chi2_o = (((data - mo) ** 2) / sig2).sum().item()
    print('star: flux', H.rel_err(got['a'], pf['a'].numpy()), 'dx', np.abs(got['dx'] - pf['dx'].numpy()).max(), 'dy',
          np.abs(got['dy'] - pf['dy'].numpy()).max(), 'chi2', abs(chi2_e.sum() - chi2_o) / chi2_o)
    assert H.rel_err(got['a'], pf['a'].numpy()) < 1e-4
    assert np.abs(got['dx'] - pf['dx'].numpy()).max() < 1e-4 and np.abs(got['dy'] - pf['dy'].numpy()).max() < 1e-4
    assert abs(chi2_e.sum() - chi2_o) / chi2_o < 1e-5
