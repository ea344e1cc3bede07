"""The cluster form of the 64 x 64 epoch kernel (csrc/joint_kernels.h, PHASE = 7; include/lcmi.h lc_joint_cluster_info):
with few epochs per GPU - a rank's share of a sharded BASELINE configs[3] fit: 25 epochs - the six phases of an epoch run
in ONE launch on several workgroups per epoch (default since round 4: six per epoch up to 32 epochs per GPU, where it pays
beside the four-launch regulariser chain; LCMI_CLUSTER=<P> forces a count, 0 switches the form off: see cluster_parts in
csrc/joint_fit.hip for what was measured), separated by flag syncs in device memory, the spectrum handed over through the XCD's L2 (plain stores,
L1-bypassing loads) when the workgroups of an epoch share an XCD, through write-through stores otherwise.  Same transforms per element as the one-workgroup kernel; the
partial sums of chi2 and of the parameter gradients are added per workgroup, then in workgroup order.

Checked here: against the one-workgroup kernel (fp32 rounding of those sums only), bitwise repeatability, odd epoch counts
(grid rounded up to groups of eight epochs), the translated / rotated paths, and the self-healing fall-back: a run whose
cluster launch gives up is redone by the library with the one-workgroup kernel, bit for bit the run that never used the
cluster form.  The reference keeps all epochs on one device (lightcurver/processes/roi_modelling.py:154-160,213) and has no
counterpart."""
import os

import numpy as np
import pytest

from lightcurver_amd.synthetic import make_roi_dataset

pytestmark = pytest.mark.gpu

FREE = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h']


def _fit(ctx, ds, M, T, env=None, lr=1e-4, rotate=False, runs=1, loss=None):
    from lightcurver_amd.joint import JointFit
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
        p = {k: np.asarray(v, dtype=np.float64) for k, v in ds['truth'].items()}
        p['a'] = 0.9 * p['a']
        if rotate:
            p['alpha'] = np.linspace(-3.0, 3.0, p['alpha'].size)
        j.set_params(**p)
        W = j.propagate_noise()
        j.set_loss(W=W, **(loss or dict(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)))
        j.set_free(FREE)
        info = []
        for _ in range(runs):
            j.run_adabelief(T, init_learning_rate=lr, schedule_learning_rate=False)
            info.append(j.cluster_info())
        out = (np.asarray(j.loss_history(), dtype=np.float64), {k: np.asarray(v, dtype=np.float64) for k, v in j.get_params().items()}, info)
        j.close()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _close(a, b, T, lr):
    ha, pa, _ = a
    hb, pb, _ = b
    assert ha.shape == hb.shape and np.all(np.isfinite(ha))
    assert np.max(np.abs(ha - hb) / np.abs(hb)) < 1e-5
    for k in FREE:
        scale = np.max(np.abs(pb[k])) + 1e-30
        tol = max(0.02 * T * lr, 3 * lr) if k == 'h' else 2e-5 * scale + 1e-7
        assert np.max(np.abs(pa[k] - pb[k])) < tol, k


@pytest.mark.parametrize('E,parts', [(6, '6'), (6, '3'), (6, '2'), (11, '5'), (25, 'auto')])
def test_cluster_launch_equals_the_one_workgroup_kernel(ctx, E, parts):
    ds = make_roi_dataset(E=E, M=2, n=64, ss=2, seed=104)
    T = 15
    a = _fit(ctx, ds, 2, T, env={'LCMI_CLUSTER': '0'})
    b = _fit(ctx, ds, 2, T, env={'LCMI_CLUSTER': parts})
    assert a[2] == [(0, 0)]
    assert b[2] == [(6 if parts == 'auto' else int(parts), 0)]   # (auto: six workgroups per epoch where they fit)
    _close(b, a, T, 1e-4)
    c = _fit(ctx, ds, 2, T, env={'LCMI_CLUSTER': parts})
    np.testing.assert_array_equal(b[0], c[0])                  # fixed summation orders: bitwise repeatable
    for k in FREE:
        np.testing.assert_array_equal(b[1][k], c[1][k])


def test_default_is_the_cluster_form_where_it_pays(ctx):
    """Without LCMI_CLUSTER in the environment: six workgroups per epoch while 6 E + 64 workgroups fit the machine's CUs (the
    chain of the second stream keeps its own), the one-workgroup kernel beyond - cluster_parts in csrc/joint_fit.hip has the
    measurements behind the rule."""
    old = os.environ.pop('LCMI_CLUSTER', None)
    try:
        for E, want in ((7, 6), (40, 0)):
            ds = make_roi_dataset(E=E, M=2, n=64, ss=2, seed=104)
            r = _fit(ctx, ds, 2, 6)
            assert r[2] == [(want, 0)], (E, r[2])
            assert np.all(np.isfinite(r[0])) and r[0][-1] < r[0][0]
    finally:
        if old is not None:
            os.environ['LCMI_CLUSTER'] = old


def test_cluster_launch_rotated_epochs(ctx):
    """Rotated epochs take the general interpolation and the ordered-gather form of T_e^T everywhere."""
    ds = make_roi_dataset(E=5, M=2, n=64, ss=2, seed=107)
    a = _fit(ctx, ds, 2, 10, env={'LCMI_CLUSTER': '0'}, rotate=True)
    b = _fit(ctx, ds, 2, 10, env={'LCMI_CLUSTER': '4'}, rotate=True)
    assert b[2] == [(4, 0)]
    _close(b, a, 10, 1e-4)


def test_a_cluster_launch_that_gives_up_is_redone_by_the_library(ctx):
    """LCMI_CLUSTER_TEST_ABORT=1 sets the abort word before the first launch: every workgroup leaves at its first barrier,
    the epoch outputs of the run are garbage.  lc_joint_run_adabelief notices at the end of the run, restores the state it
    copied at the start and runs the same iterations with the one-workgroup kernel: the numbers of a fit that never used
    the cluster form, and the object stays with that kernel."""
    ds = make_roi_dataset(E=6, M=2, n=64, ss=2, seed=104)
    T = 12
    a = _fit(ctx, ds, 2, T, env={'LCMI_CLUSTER': '0'}, runs=2)
    b = _fit(ctx, ds, 2, T, env={'LCMI_CLUSTER': '6', 'LCMI_CLUSTER_TEST_ABORT': '1'}, runs=2)
    assert b[2] == [(0, 1), (0, 1)]
    np.testing.assert_array_equal(a[0], b[0])
    for k in FREE:
        np.testing.assert_array_equal(a[1][k], b[1][k])


@pytest.mark.parametrize('loss', [None, dict(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0),
                                  dict(lam_positivity=100.0, lam_pts_source=0.01)])
@pytest.mark.parametrize('cluster', ['0', '6'])
def test_one_launch_regulariser_chain_equals_the_launches(ctx, loss, cluster):
    """The regulariser of the 128 x 128 background grid as ONE launch (csrc/joint_reg_mfma.h, mreg_chain_kernel: the stages
    of the launch chain separated by syncs over its 64 resident workgroups) (LCMI_REG_CHAIN=1) against the eight launches (the default):
    same stages, same summation orders - identical bits, beside the one-workgroup epoch kernel and beside the cluster form."""
    ds = make_roi_dataset(E=6, M=2, n=64, ss=2, seed=104)
    T = 12
    # (LCMI_REG_FUSED=0: the eight launches the one-launch form restates; the default since round 4 is the four-launch form of
    #  csrc/joint_reg_fused.h, which adds its values in another order - tests/test_joint_paths_gpu.py)
    a = _fit(ctx, ds, 2, T, env={'LCMI_CLUSTER': cluster, 'LCMI_REG_CHAIN': '0', 'LCMI_REG_FUSED': '0'}, loss=loss)
    b = _fit(ctx, ds, 2, T, env={'LCMI_CLUSTER': cluster, 'LCMI_REG_CHAIN': '1', 'LCMI_REG_FUSED': '0'}, loss=loss)
    np.testing.assert_array_equal(a[0], b[0])
    for k in FREE:
        np.testing.assert_array_equal(a[1][k], b[1][k])
    assert a[0][-1] < a[0][0]
