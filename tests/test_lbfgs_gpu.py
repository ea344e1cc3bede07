"""The bounded L-BFGS with its vectors on the device (csrc/joint_lbfgs.h, lc_joint_run_lbfgs; `north_star`: "L-BFGS
parameter updates fused on-device") against scipy's L-BFGS-B on the float64 oracle loss - what STARRED's
Optimizer(method='l-bfgs-b') runs for the reference at roi_modelling.py:278-280 (translations + fluxes stage of the ROI
fit).  Iterates differ by construction (another line search); the optimum must be the same."""
import os

import numpy as np
import pytest

from oracle import model as om, optim as oo
from lightcurver_amd.synthetic import make_roi_dataset
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _problem(ctx, E=8, M=2, n=16, ss=2, seed=61):
    from lightcurver_amd.joint import JointFit
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=seed)
    rng = np.random.default_rng(seed + 1)
    p = {k: np.array(v, dtype=np.float64) for k, v in ds['truth'].items()}
    p['a'] = p['a'] * rng.uniform(0.7, 1.3, p['a'].shape)
    p['dx'] = p['dx'] + rng.normal(0, 0.3, E)
    p['dy'] = p['dy'] + rng.normal(0, 0.3, E)
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, ctx)
    j.set_params(**p)
    return ds, p, j


def test_device_lbfgs_reaches_the_scipy_optimum(ctx):
    E, M, n, ss = 8, 2, 16, 2
    ds, p, j = _problem(ctx, E, M, n, ss)
    free = ['a', 'dx', 'dy']
    j.set_loss(lam_flux_uniformity=1.0)          # roi_modelling.py:273-276: only the flux-scatter term in this stage
    j.set_free(free)
    lower = dict(a=np.zeros(E * M), dx=np.full(E, -n / 2), dy=np.full(E, -n / 2))
    upper = dict(a=np.full(E * M, 1e10), dx=np.full(E, n / 2), dy=np.full(E, n / 2))
    hist, nit, nev = j.run_lbfgs(300, lower, upper)
    got = j.get_params()
    assert nit >= 5 and nev >= nit and np.all(np.diff(hist) <= 0)    # monotone: every accepted step satisfies Armijo
    po = {k: om.T(v) for k, v in p.items()}
    data, sig2, psf = om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, lam_fu=1.0)
    pf, lh, res = oo.lbfgsb(fn, po, free, 300, {k: (lower[k], upper[k]) for k in free})
    print('device L-BFGS', nit, 'iterations', nev, 'evaluations, loss', hist[-1], '| scipy', res.nit, res.nfev, res.fun)
    assert abs(hist[-1] - res.fun) / res.fun < 1e-4
    assert hist[-1] <= res.fun * (1 + 1e-4)
    assert H.rel_err(got['a'], pf['a'].numpy()) < 2e-3
    assert np.abs(got['dx'] - pf['dx'].numpy()).max() < 5e-3 and np.abs(got['dy'] - pf['dy'].numpy()).max() < 5e-3
    # the loss the device reports is the loss of the parameters it leaves behind
    loss_now, _ = j.loss_grad(free)
    assert abs(loss_now - hist[-1]) <= 1e-6 * abs(loss_now)


def test_bounds_are_respected_and_facade_takes_the_device_path(ctx):
    from copy import deepcopy
    from lightcurver_amd.starred.deconvolution.deconvolution import setup_model
    from lightcurver_amd.starred.deconvolution.loss import Loss
    from lightcurver_amd.starred.deconvolution.parameters import ParametersDeconv
    from lightcurver_amd.starred.optim.optimization import Optimizer
    E, M, n, ss = 6, 1, 16, 2
    ds, p, j = _problem(ctx, E, M, n, ss, seed=62)
    free = ['a', 'dx', 'dy']
    j.set_loss()
    j.set_free(free)
    # an upper bound below the optimum of the fluxes: they must end ON the bound
    cap = 0.5 * np.asarray(ds['truth']['a'])
    hist, nit, nev = j.run_lbfgs(100, dict(a=np.zeros(E)), dict(a=cap))
    a = j.get_params()['a']
    assert np.all(a <= cap * (1 + 1e-6)) and np.allclose(a, cap, rtol=1e-5)
    j.close()
    # facade: Optimizer(method='l-bfgs-b') runs the device optimiser; LCMI_LBFGS_SCIPY=1 the host cross-check
    data, noise = ds['data'].astype(np.float64), ds['noisemap'].astype(np.float64)
    out = {}
    for env in ('', '1'):
        model, k_init, k_up, k_down, _ = setup_model(data, noise ** 2, ds['psf'], np.array([0.]), np.array([0.]), ss,
                                                     list(0.8 * np.asarray(ds['truth']['a'])))
        fixed = deepcopy(k_init)
        for name in free:
            del fixed['kwargs_analytic'][name]
        pars = ParametersDeconv(kwargs_init=k_init, kwargs_fixed=fixed, kwargs_up=k_up, kwargs_down=k_down)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            loss = Loss(data, model, pars, noise ** 2)
        opt = Optimizer(loss, pars, method='l-bfgs-b')
        if env:
            os.environ['LCMI_LBFGS_SCIPY'] = env
        try:
            best, logL, extra, _ = opt.minimize(maxiter=200)
        finally:
            os.environ.pop('LCMI_LBFGS_SCIPY', None)
        out[env] = (np.asarray(best), -logL, extra)
    assert 'evaluations' in out[''][2] and 'scipy_result' in out['1'][2]
    assert abs(out[''][1] - out['1'][1]) / out['1'][1] < 1e-4
    assert np.abs(out[''][0] - out['1'][0]).max() < 5e-3 * max(np.abs(out['1'][0]).max(), 1.0)


def test_recycled_work_space_does_not_reach_the_results(ctx, monkeypatch):
    """lc_joint_run_lbfgs allocates its vectors per call; hipMalloc returns recycled bytes.  With the work space filled with
    0xFF (NaN patterns, LCMI_LBFGS_POISON) instead of zeros, an unbounded fit with the background free - the case in which
    a NaN read before the first write would move the starting point - must give the same iterates bit for bit."""
    E, M, n, ss = 6, 2, 16, 2
    out = []
    for poison in (False, True):
        ds, p, j = _problem(ctx, E, M, n, ss, seed=63)
        j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0)
        j.set_free(['a', 'dx', 'dy', 'h'])
        if poison:
            monkeypatch.setenv('LCMI_LBFGS_POISON', '1')
        hist, nit, nev = j.run_lbfgs(15)                      # no bounds at all
        monkeypatch.delenv('LCMI_LBFGS_POISON', raising=False)
        got = j.get_params()
        assert np.all(np.isfinite(hist)) and all(np.all(np.isfinite(v)) for v in got.values())
        out.append((hist, nit, nev, got))
        j.close()
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
    assert np.array_equal(out[0][0], out[1][0])
    for k in out[0][3]:
        assert np.array_equal(out[0][3][k], out[1][3][k]), k
