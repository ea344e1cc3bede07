"""Edge cases of the C ABI: the largest star / source counts the kernels accept, everything beyond them, and
malformed arguments.  Numerical checks against the oracle use the tolerances of test_psf_gpu.py / test_joint_gpu.py."""
import ctypes as C

import numpy as np
import pytest

from oracle import model as om, optim as oo
from lightcurver_amd import _lib
from lightcurver_amd.synthetic import make_psf_dataset, make_roi_dataset
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_psf_sixteen_stars_per_frame(ctx):
    """S = 16 is the most stars a frame may hold (PsfCfg::MAXS)."""
    from lightcurver_amd.psf_batch import PsfBatch
    F, S, n, ss = 2, 16, 16, 2
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=77)
    rng = np.random.default_rng(3)
    plist = [H.psf_initial_params(ds, f, ss, rng, 0.2) for f in range(F)]
    b = PsfBatch(ds['data'], H.weights_from(ds), ss, ctx)
    b.set_moffat(H.moffat_array(plist))
    b.set_stars(H.stars_array(plist))
    b.set_grid(np.stack([p['B'].numpy() for p in plist]))
    b.set_regularization(None, 1.0, 1.0)
    out = b.evaluate()
    for f in range(F):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, lam_scales=1.0, lam_hf=1.0)
        L, g = oo.value_and_grad(fn, plist[f], ['B', 'a', 'x0', 'y0'])
        assert abs(out['loss'][f] - float(L)) / float(L) < 2e-5
        assert H.rel_err(out['grad_grid'][f].ravel(), g['B'].numpy().ravel()) < 5e-5
        assert H.rel_err(out['grad_stars'][f][:, 0], g['a'].numpy()) < 5e-5
    with pytest.raises(_lib.LcError):
        big = make_psf_dataset(F=1, S=17, n=16, ss=2, seed=1)
        PsfBatch(big['data'], H.weights_from(big), 2, ctx)


def test_joint_eight_point_sources(ctx):
    """M = 8 is the most point sources of a joint fit (kMaxSources); M = 0 (background only) is legal too."""
    from lightcurver_amd.joint import JointFit
    E, M, n, ss = 3, 8, 16, 2
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=5)
    p = {k: np.array(v, dtype=np.float64) for k, v in ds['truth'].items()}
    p['h'] = p['h'] + 1e-3 * np.random.default_rng(1).standard_normal(p['h'].shape)
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, ctx)
    j.set_params(**p)
    j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_flux_uniformity=0.3)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean']
    j.set_free(free)
    po = {k: om.T(v) for k, v in p.items()}
    data, sig2, psf = om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])
    L, g = oo.value_and_grad(lambda q: om.deconv_loss(q, data, sig2, psf, ss, lam_scales=1.0, lam_hf=1.0, lam_fu=0.3), po, free)
    loss, grads = j.loss_grad(free)
    assert abs(loss - float(L)) / float(L) < 3e-5
    for k in free:
        assert H.rel_err(grads[k], g[k].numpy()) < 1e-4, k
    with pytest.raises(_lib.LcError):
        JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, 9, ctx)
    # background-only model
    j0 = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, 0, ctx)
    j0.set_params(h=p['h'], dx=p['dx'], dy=p['dy'], alpha=p['alpha'], mean=p['mean'])
    j0.set_free(['h', 'mean'])
    p0 = dict(po)
    p0['a'] = om.T(np.zeros(0))
    p0['c_x'] = om.T(np.zeros(0))
    p0['c_y'] = om.T(np.zeros(0))
    L0, g0 = oo.value_and_grad(lambda q: om.deconv_loss(q, data, sig2, psf, ss), p0, ['h', 'mean'])
    l0, gr0 = j0.loss_grad(['h', 'mean'])
    assert abs(l0 - float(L0)) / float(L0) < 3e-5
    assert H.rel_err(gr0['h'], g0['h'].numpy()) < 1e-4


def test_malformed_arguments_are_refused(ctx):
    lib = _lib.lib()
    h = C.c_void_p()
    d = np.zeros((2, 2, 20, 20), np.float32)
    # 20 x 20 stamps: no kernel instantiated
    assert lib.lc_psf_supported(20, 2) == 0 and lib.lc_joint_supported(20, 2) == 0
    assert lib.lc_psf_batch_create(ctx.h, 2, 2, 20, 2, _lib.ptr(d), _lib.ptr(d), C.byref(h)) == -3
    assert b'stamp size' in lib.lc_last_error(ctx.h)
    # null data pointer, zero frames
    assert lib.lc_psf_batch_create(ctx.h, 2, 2, 16, 2, None, _lib.ptr(d), C.byref(h)) == -1
    assert lib.lc_psf_batch_create(ctx.h, 0, 2, 16, 2, _lib.ptr(d), _lib.ptr(d), C.byref(h)) == -1
    assert lib.lc_joint_create(ctx.h, 0, 1, 16, 2, _lib.ptr(d), _lib.ptr(d), _lib.ptr(d), C.byref(h)) == -1
    # wrong element count for a parameter block; alpha can never be freed
    from lightcurver_amd.joint import JointFit
    ds = make_roi_dataset(E=2, M=1, n=16, ss=2, seed=2)
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, 1, ctx)
    v = np.zeros(5, np.float32)
    assert lib.lc_joint_set_param(j.h, 0, _lib.ptr(v), 5) == -1
    mask = (C.c_int32 * 8)(0, 0, 0, 0, 0, 1, 0, 0)
    assert lib.lc_joint_set_free(j.h, mask) == -3
    # non-finite data and non-positive variances are masked out at creation, not propagated
    data = ds['data'].copy()
    data[0, 3, 3] = np.nan
    s2 = ds['noisemap'].astype(np.float64) ** 2
    s2[1, 4, 4] = 0.0
    jn = JointFit(data, s2, ds['psf'], 2, 1, ctx)
    jn.set_params(**{k: np.asarray(v) for k, v in ds['truth'].items()})
    model, chi2 = jn.model()
    assert np.all(np.isfinite(model)) and np.all(np.isfinite(chi2))
