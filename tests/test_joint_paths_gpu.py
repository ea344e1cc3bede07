"""The joint fit has two implementations of two of its steps, chosen per object at creation / per iteration:

* the starlet regulariser of the large background grids (N >= 128): two-sided products on the fp32 matrix cores
  (csrc/joint_reg_mfma.h, default) or the a-trous cascade kernels (LCMI_REG_CASCADE=1);
* inside lc_joint_run_adabelief with the background free: reduction over the epochs and update in one launch (default) or
  as two kernels (LCMI_SPLIT_UPDATE=1; the kernels the sharded drive uses); the fused launch learns that the second
  stream's regulariser is complete from a flag it checks itself (default) or from an event wait (LCMI_EVENT_SYNC=1).

Both pairs compute the same numbers in different summation orders; these tests run the same fit through each and compare
(fp32 rounding of the sums only: 1e-5 relative on the loss history, a few 1e-6 of the parameter scale on the parameters).
The C5 problem at its full epoch count (1000 x 128 x 128, 4 sources; BASELINE.json configs[4]) runs here too: no oracle at
that size, so the checks are size-independent ones (decreasing loss, determinism, epochs independent of their
position in the batch)."""
import os

import numpy as np
import pytest

from lightcurver_amd.synthetic import make_roi_dataset

pytestmark = pytest.mark.gpu


def _fit(ctx, ds, M, T, env=None, lr=1e-4, idx=None):
    from lightcurver_amd.joint import JointFit
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        sel = slice(None) if idx is None else idx
        E = ds['data'].shape[0]
        j = JointFit(ds['data'][sel], ds['noisemap'][sel].astype(np.float64) ** 2, ds['psf'][sel], 2, M, ctx)
        p = {k: np.asarray(v, dtype=np.float64) for k, v in ds['truth'].items()}
        if idx is not None:
            p['a'] = p['a'].reshape(E, M)[idx].reshape(-1)
            for k in ('dx', 'dy', 'alpha', 'mean'):
                p[k] = p[k][idx]
        p['a'] = 0.9 * p['a']
        j.set_params(**p)
        W = j.propagate_noise()
        j.set_loss(W=W, lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
        j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
        j.run_adabelief(T, init_learning_rate=lr, schedule_learning_rate=False)
        out = (np.asarray(j.loss_history(), dtype=np.float64), {k: np.asarray(v, dtype=np.float64) for k, v in j.get_params().items()})
        j.close()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _compare(a, b, names, T, lr):
    ha, pa = a
    hb, pb = b
    assert ha.shape == hb.shape
    assert np.max(np.abs(ha - hb) / np.abs(hb)) < 1e-5
    for k in names:
        scale = np.max(np.abs(pb[k])) + 1e-30
        tol = 2e-5 * scale + 1e-7
        # a background pixel whose starlet coefficient is within rounding of zero may take the other sign of the l1
        # sub-gradient in one of the two forms: AdaBelief steps of size <= lr in opposite directions in that iteration
        # (2 lr apart), a small fraction of the T * lr a pixel can move at all
        if k == 'h':
            tol = max(0.02 * T * lr, 3 * lr)
        assert np.max(np.abs(pa[k] - pb[k])) < tol, k


@pytest.mark.parametrize('E,n,M', [(8, 64, 2), (3, 128, 4)])
def test_matrix_core_regulariser_equals_the_cascade(ctx, E, n, M):
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
    a = _fit(ctx, ds, M, 25)
    b = _fit(ctx, ds, M, 25, env={'LCMI_REG_CASCADE': '1'})
    _compare(a, b, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'), 25, 1e-4)
    assert a[0][-1] < a[0][0]


@pytest.mark.parametrize('E,n,M', [(8, 64, 2), (3, 128, 4)])
def test_regulariser_chain_forms_agree(ctx, E, n, M):
    """The matrix-core regulariser of the large grids exists as batched tiled products over the scales (default; the K loop of
    a tile only covers the band of the cumulative smoothing operator), the same with all K slices (LCMI_REG_DENSE=1: the
    skipped slices multiply by zeros, so the bits are the same) and as the block-column kernels of round 2
    (LCMI_REG_MFMA_V1=1: another summation order)."""
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
    a = _fit(ctx, ds, M, 20)
    b = _fit(ctx, ds, M, 20, env={'LCMI_REG_DENSE': '1'})
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1]['h'], b[1]['h'])
    c = _fit(ctx, ds, M, 20, env={'LCMI_REG_MFMA_V1': '1'})
    _compare(a, c, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'), 20, 1e-4)


@pytest.mark.parametrize('E,n,M', [(8, 64, 2), (3, 128, 4), (6, 64, 1)])
def test_four_launch_chain_equals_the_eight_launch_one(ctx, E, n, M):
    """Default since round 4 (csrc/joint_reg_fused.h): the element-wise launches of the batched-product chain folded into the
    products' operand fetches and epilogues, the sums left to the fused reduction + update.  Element by element the same
    arithmetic as the eight launches (LCMI_REG_FUSED=0); the values of the terms and the inner products of the point-source
    term are added per 64 x 64 tile instead of per 256-pixel block: fp32 rounding of those sums only.  With the split update
    (LCMI_SPLIT_UPDATE=1) the same chain hands greg / regs over through one more launch: the same bits as the fused form."""
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
    a = _fit(ctx, ds, M, 25)
    b = _fit(ctx, ds, M, 25, env={'LCMI_REG_FUSED': '0'})
    _compare(a, b, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'), 25, 1e-4)
    assert a[0][-1] < a[0][0]
    c = _fit(ctx, ds, M, 25, env={'LCMI_SPLIT_UPDATE': '1'})
    np.testing.assert_array_equal(a[0], c[0])
    np.testing.assert_array_equal(a[1]['h'], c[1]['h'])
    # one iteration: identical h gradient (same products, same order of the planes' sum), so identical h after one step
    a1 = _fit(ctx, ds, M, 1)
    b1 = _fit(ctx, ds, M, 1, env={'LCMI_REG_FUSED': '0'})
    np.testing.assert_array_equal(a1[1]['h'], b1[1]['h'])


@pytest.mark.parametrize('E,n,M', [(8, 64, 2), (5, 32, 2), (3, 128, 4)])
def test_fused_reduction_and_update_equals_the_two_kernels(ctx, E, n, M):
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
    a = _fit(ctx, ds, M, 25)
    b = _fit(ctx, ds, M, 25, env={'LCMI_SPLIT_UPDATE': '1'})
    # same reduction order in both forms: the histories are identical, not merely close
    np.testing.assert_array_equal(a[0], b[0])
    for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'):
        np.testing.assert_array_equal(a[1][k], b[1][k])


def test_flag_synchronised_update_equals_the_event_synchronised_one(ctx):
    """Device loop, N >= 128: the fused update checks the completion flag of the regulariser chain in the kernel (default)
    or the host enqueues a cross-stream event wait in front of it (LCMI_EVENT_SYNC=1).  Same kernels, same numbers."""
    ds = make_roi_dataset(E=8, M=2, n=64, ss=2, seed=104)
    a = _fit(ctx, ds, 2, 40)
    b = _fit(ctx, ds, 2, 40, env={'LCMI_EVENT_SYNC': '1'})
    np.testing.assert_array_equal(a[0], b[0])
    for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'):
        np.testing.assert_array_equal(a[1][k], b[1][k])


@pytest.mark.parametrize('E,n,M', [(8, 64, 2), (40, 64, 2), (3, 128, 4), (5, 32, 2)])
def test_loop_synchronisation_forms_give_the_same_bits(ctx, E, n, M):
    """How the two streams of the device loop tell each other where they are changes no arithmetic: the regulariser chain of
    an iteration starts behind a gate kernel that the next epoch launch opens (default) or behind an event recorded after the
    update (LCMI_UPD_EVENT=1); the update learns that the chain is complete from an extra block of the epoch launch that waits
    for it (default where the chain is the shorter path) or from its own poll / an event wait (LCMI_EPOCH_WAIT_OFF=1); all of
    it as events (LCMI_EVENT_SYNC=1).  Identical histories and parameters."""
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
    a = _fit(ctx, ds, M, 20)
    for env in ({'LCMI_UPD_EVENT': '1'}, {'LCMI_EPOCH_WAIT_OFF': '1'}, {'LCMI_EVENT_SYNC': '1'}):
        b = _fit(ctx, ds, M, 20, env=env)
        np.testing.assert_array_equal(a[0], b[0], err_msg=str(env))
        for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'):
            np.testing.assert_array_equal(a[1][k], b[1][k], err_msg=str(env) + k)
    assert a[0][-1] < a[0][0]


@pytest.mark.parametrize('E,n,M,free', [(7, 32, 1, ('a', 'dx', 'dy', 'mean')), (5, 16, 2, ('a', 'mean')), (3, 64, 1, ('a', 'dx', 'dy'))])
def test_persistent_star_photometry_loop_equals_the_launch_per_iteration_one(ctx, E, n, M, free):
    """Photometry at fixed positions (no background, only per-epoch parameters free, nothing coupling the epochs): the whole
    AdaBelief loop runs inside one launch of the point-source-only kernel (default) or as one kernel pair per iteration
    (LCMI_PS_LOOP=1; also the form the reference's default star photometry takes, whose shared position is free).
    Same filters, reductions and update per epoch: identical parameters; the loss history differs only in the order in
    which the epochs are added."""
    from lightcurver_amd.joint import JointFit
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=31, with_background=False)
    out = []
    for env in ({}, {'LCMI_PS_LOOP': '1'}):
        os.environ.update(env)
        try:
            j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
            p = {k: np.asarray(v, dtype=np.float64) for k, v in ds['truth'].items()}
            p['a'] = 0.8 * p['a']
            p['h'] = np.zeros_like(p['h'])
            j.set_params(**p)
            j.set_loss(lam_positivity_ps=3.0)
            j.set_free(list(free))
            j.run_adabelief(60, init_learning_rate=1e-3, schedule_learning_rate=True)
            j.run_adabelief(40, init_learning_rate=1e-3, schedule_learning_rate=True)   # a second call continues the first
            out.append((np.asarray(j.loss_history(), dtype=np.float64), {k: np.asarray(v) for k, v in j.get_params().items()}))
            j.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    (ha, pa), (hb, pb) = out
    assert ha.shape == hb.shape == (101,)
    assert np.max(np.abs(ha - hb) / np.abs(hb)) < 2e-6 and ha[-1] < ha[0]
    for k in ('a', 'dx', 'dy', 'mean'):
        np.testing.assert_array_equal(pa[k], pb[k])


def test_tiled_column_passes_equal_the_direct_ones(ctx):
    """128 x 128 ROIs keep the spectrum of an epoch in global memory; from 96 epochs up the column passes stage it through
    an LDS tile (LCMI_TILE_COLS forces either form).  Only the data movement differs; the two are separate kernel builds,
    so the compiler's choice of fused multiply-adds may differ in the last bit."""
    ds = make_roi_dataset(E=3, M=4, n=128, ss=2, seed=104)
    a = _fit(ctx, ds, 4, 12, env={'LCMI_TILE_COLS': '0'})
    b = _fit(ctx, ds, 4, 12, env={'LCMI_TILE_COLS': '1'})
    assert np.max(np.abs(a[0] - b[0]) / np.abs(b[0])) < 1e-6
    _compare(a, b, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'), 12, 1e-4)


@pytest.mark.parametrize('E,parts', [(3, None), (3, '1'), (5, '2')])
def test_stencil_in_the_reduction_equals_the_slab_form(ctx, E, parts):
    """128 x 128 ROIs, translated epochs, device loop: the reduction over the epochs applies the adjoint interpolation
    stencil T_e^T itself, reading the scene-gradient rows from the spectrum scratch (no phase D, no per-epoch
    slabs: LCMI_STENCIL_REDUCE=1 - measured no faster, so not the default), or phase D writes one slab per epoch that the
    reduction adds up.  Same taps and weights; the epochs are added in another order."""
    ds = make_roi_dataset(E=E, M=4, n=128, ss=2, seed=106)
    env = {} if parts is None else {'LCMI_EPOCH_PARTS': parts}
    a = _fit(ctx, ds, 4, 12, env=env)
    b = _fit(ctx, ds, 4, 12, env=dict(env, LCMI_STENCIL_REDUCE='1'))
    assert np.max(np.abs(a[0] - b[0]) / np.abs(b[0])) < 1e-6
    _compare(a, b, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'), 12, 1e-4)


@pytest.mark.parametrize('E,n,M', [(8, 64, 2), (3, 128, 4)])
def test_point_source_term_beside_the_chain_equals_the_batched_form(ctx, E, n, M):
    """The point-source starlet term as tiles in one launch on a third stream beside the regulariser chain (LCMI_PTS_SIDE=1;
    measured slower, opt-in) against the same term as one more product in each batch of the chain (default): other summation
    order of the scale-0 stencil, same numbers to fp32 rounding."""
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
    # (LCMI_EVENT_SYNC=1: this opt-in form joins a THIRD stream into the chain with events; in a process that has created many
    #  streams before, the runtime may put two of the three onto one hardware queue, and the update's in-kernel wait for a chain
    #  that is held up there ran out once in a full-suite run.  What is compared here is the arithmetic of the two forms.)
    a = _fit(ctx, ds, M, 25, env={'LCMI_PTS_SIDE': '1', 'LCMI_EVENT_SYNC': '1'})
    b = _fit(ctx, ds, M, 25)
    _compare(a, b, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'), 25, 1e-4)


@pytest.mark.parametrize('parts', ['2', '3'])
def test_row_block_regulariser_equals_the_batched_products(ctx, parts):
    """The regulariser of the 128 x 128 grid cut by rows (csrc/joint_reg_rows.h, LCMI_REG_ROWS=1: forward products, S rows and
    adjoint products of a row block in ONE workgroup, 16 x 16 x 4 fp32 MFMA tiles with a permuted k order; the point-source
    term as tiles) against the batched-product chain (the default) and against the cascade: same mathematics, other
    summation orders."""
    ds = make_roi_dataset(E=8, M=2, n=64, ss=2, seed=104)
    a = _fit(ctx, ds, 2, 25)
    b = _fit(ctx, ds, 2, 25, env={'LCMI_REG_ROWS': '1', 'LCMI_REG_ROWS_PARTS': parts})
    c = _fit(ctx, ds, 2, 25, env={'LCMI_REG_CASCADE': '1'})
    _compare(b, a, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'), 25, 1e-4)
    _compare(b, c, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'), 25, 1e-4)
    assert b[0][-1] < b[0][0]


@pytest.mark.parametrize('parts', ['2', '4'])
def test_epoch_spread_over_workgroups_equals_the_one_workgroup_kernel(ctx, parts):
    """128 x 128 ROIs with fewer epochs than CUs: the six phases of an epoch become six launches on a grid
    (epochs, workgroups per epoch) with the spectrum in global memory between them (LCMI_EPOCH_PARTS forces the count;
    1 = the single kernel).  Same arithmetic per element; only the partial sums of chi2 and of the parameter gradients
    are added in a different order."""
    ds = make_roi_dataset(E=3, M=4, n=128, ss=2, seed=105)
    a = _fit(ctx, ds, 4, 12, env={'LCMI_EPOCH_PARTS': '1'})
    b = _fit(ctx, ds, 4, 12, env={'LCMI_EPOCH_PARTS': parts})
    assert np.max(np.abs(a[0] - b[0]) / np.abs(b[0])) < 1e-6
    _compare(a, b, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'), 12, 1e-4)


@pytest.mark.parametrize('parts', [None, '8', '4', '1'])
def test_n64_fit_spread_over_workgroups_equals_the_one_workgroup_kernel(ctx, parts):
    """64 x 64 ROIs (BASELINE.json configs[3]) with few epochs per GPU: the device loop through the phased launches of the
    split form (LCMI_N128_SPLIT=1) against the one-workgroup LDS-spectrum kernel (the default).  Separate kernel builds,
    partial sums added in another order."""
    ds = make_roi_dataset(E=6, M=2, n=64, ss=2, seed=104)
    env = {'LCMI_N128_SPLIT': '1'}
    if parts:
        env['LCMI_EPOCH_PARTS'] = parts
    a = _fit(ctx, ds, 2, 15, env={'LCMI_N128_SPLIT': '0'})
    b = _fit(ctx, ds, 2, 15, env=env)
    assert np.max(np.abs(a[0] - b[0]) / np.abs(b[0])) < 2e-6
    _compare(a, b, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'), 15, 1e-4)


def test_update_waits_for_a_regulariser_chain_that_runs_late(ctx):
    """The fused update reads the regulariser's completion flag in the kernel.  Normally the chain is done before the
    epoch kernel ends and nothing waits; LCMI_REG_DELAY_US holds the second stream back by 300 us per iteration (five epoch
    kernels), so every update of this fit really waits in the kernel - with its blocks resident - and the chain must
    still find room to run.  Same numbers as the undelayed fit, and no time-out."""
    ds = make_roi_dataset(E=8, M=2, n=64, ss=2, seed=104)
    a = _fit(ctx, ds, 2, 30)
    b = _fit(ctx, ds, 2, 30, env={'LCMI_REG_DELAY_US': '300'})
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1]['h'], b[1]['h'])


def test_c5_at_its_full_epoch_count(ctx):
    """BASELINE.json configs[4] on one GPU: 1000 epochs x 128 x 128, 4 point sources + background, everything free."""
    E, M, n, T = 1000, 4, 128, 6
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=105)
    h, p = _fit(ctx, ds, M, T)
    assert np.all(np.isfinite(h)) and h[-1] < h[0] - 1e-3 * abs(h[0])   # (AdaBelief's first steps are not monotone)
    h2, p2 = _fit(ctx, ds, M, T)
    np.testing.assert_array_equal(h, h2)          # fixed summation orders: bitwise repeatable
    np.testing.assert_array_equal(p['h'], p2['h'])
    # the chi2 part is a sum over independent epochs: a fit of the first 125 epochs alone starts from the same per-epoch
    # numbers (first loss = its share), and its per-epoch parameters after one iteration equal those inside the full fit
    hs, ps = _fit(ctx, ds, M, 1, idx=np.arange(125))
    hf, pf = _fit(ctx, ds, M, 1)
    for k in ('dx', 'dy', 'mean'):
        assert np.max(np.abs(ps[k] - pf[k][:125])) < 1e-6 * (np.max(np.abs(pf[k])) + 1e-3), k


@pytest.mark.parametrize('n,E,free', [(32, 5, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h')), (64, 4, ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h')),
                                      (16, 6, ('a', 'dx', 'dy'))])
def test_device_resident_parameter_history(ctx, n, E, free):
    """return_param_history (star_photometry.py:119, roi_modelling.py:331): the update kernels - single-workgroup, multi-block,
    fused reduction + update, and the persistent point-source loop - store the free blocks after every update in a
    device-resident history; its rows equal what get_params() returns after each iteration of a one-iteration-per-call
    drive of the same fit."""
    from lightcurver_amd.joint import JointFit
    M, T = 2, 9
    with_h = 'h' in free
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104, with_background=with_h)
    rows = []
    for history in (True, False):
        j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
        p = {k: np.asarray(v, dtype=np.float64) for k, v in ds['truth'].items()}
        p['a'] = 0.9 * p['a']
        j.set_params(**p)
        if with_h:
            j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
        else:
            j.set_loss(lam_positivity_ps=2.0)
        j.set_free(list(free))
        cfg = dict(init_learning_rate=1e-4, schedule_learning_rate=False)
        if history:
            P = j.param_history_begin(T)
            assert P == sum(j.sizes[k] for k in free)
            j.run_adabelief(T - 4, **cfg)
            j.run_adabelief(4, **cfg)          # a second call appends
            rows.append(j.param_history())
            j.param_history_end()
        else:
            ref = []
            for _ in range(T):
                j.run_adabelief(1, **cfg)
                got = j.get_params()
                ref.append(np.concatenate([got[k] for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'alpha', 'h', 'mean') if k in free]))
            rows.append(np.asarray(ref))
        j.close()
    assert rows[0].shape == rows[1].shape and rows[0].shape[0] == T
    np.testing.assert_array_equal(rows[0], rows[1])


@pytest.mark.parametrize('E,M,n', [(6, 2, 64), (5, 3, 32), (4, 2, 128)])
def test_point_source_term_behind_the_all_reduce_in_one_launch(ctx, E, M, n):
    """The step-by-step / sharded drive evaluates the point-source starlet term behind the all-reduce (it needs the mean
    fluxes of ALL ranks): the background part stays with the chain on the second stream, the term itself is one launch
    (gm_pts_direct_kernel: Pbar, scale-0 starlet, l1 sub-gradient, exact adjoint and the inner products on 16 x 16 tiles).
    Against the twelve-launch form of the same term (LCMI_PTS_CHAIN=1) and against the device loop of one GPU, which
    evaluates it with the matrix-core chain: same arithmetic in other orders, so equal to rounding."""
    import os
    from lightcurver_amd.joint import JointFit
    from lightcurver_amd.synthetic import make_roi_dataset
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=300 + n)
    T = 12

    def run(stepwise, env=None):
        os.environ.update(env or {})
        try:
            j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
            p = dict(ds['truth'])
            p['a'] = np.asarray(p['a']) * 0.9
            j.set_params(**p)
            j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.05, lam_flux_uniformity=10.0)
            j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
            if stepwise:
                for _ in range(T):
                    j.step_local()
                    j.step_update(init_learning_rate=1e-3, schedule_learning_rate=False)
            else:
                j.run_adabelief(T, init_learning_rate=1e-3, schedule_learning_rate=False)
            out = (np.asarray(j.loss_history(), np.float64), j.get_params())
            j.close()
            return out
        finally:
            for k in (env or {}):
                os.environ.pop(k, None)

    # default since round 4 at N >= 128: the term from separable tables INSIDE the update launch (PtsTail, csrc/joint_gm.h);
    # LCMI_PTS_TAIL=0: the one-launch tile kernel in front of the update
    h_tail, p_tail = run(True)
    h_direct, p_direct = run(True, {'LCMI_PTS_TAIL': '0'})
    h_chain, p_chain = run(True, {'LCMI_PTS_CHAIN': '1'})
    h_loop, p_loop = run(False)
    scale = np.abs(h_loop).max()
    assert np.abs(h_direct - h_chain).max() <= 2e-6 * scale
    assert np.abs(h_tail - h_direct).max() <= 5e-6 * scale
    for k in ('a', 'c_x', 'c_y', 'dx', 'dy'):
        ref = max(np.abs(p_loop[k]).max(), 1e-3)
        assert np.abs(p_tail[k] - p_direct[k]).max() <= 2e-5 * ref, k
    assert np.abs(h_direct - h_loop).max() <= 2e-5 * scale
    for k in ('a', 'c_x', 'c_y', 'dx', 'dy'):
        ref = max(np.abs(p_loop[k]).max(), 1e-3)
        assert np.abs(p_direct[k] - p_chain[k]).max() <= 2e-5 * ref, k
        assert np.abs(p_direct[k] - p_loop[k]).max() <= 2e-4 * ref, k


def test_device_loop_on_one_hardware_queue(ctx):
    """The runtime has a few hardware queues (four by default) and maps further streams onto the same ones.  With the two streams
    of a fit in ONE queue a gate kernel on the second stream would wait for an epoch launch that cannot start before the gate
    has ended; the library probes once per object whether its streams run side by side (probe_streams, csrc/joint_fit.hip) and
    keeps the events where they do not.  GPU_MAX_HW_QUEUES=1 forces that case (it must be set before HIP starts: a process of
    its own): same numbers as the default run, no time-out."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tools', 'queue_sharing.py')
    outs = {}
    for q in ('1', None):
        env = dict(os.environ)
        env.pop('GPU_MAX_HW_QUEUES', None)
        if q:
            env['GPU_MAX_HW_QUEUES'] = q
        r = subprocess.run([sys.executable, tool, '1'], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[q] = (r.stdout, r.stderr)
    assert 'NO (shared hardware queue)' in outs['1'][1]
    assert 'beside the main stream: yes' in outs[None][1]
    loss = {q: [ln.split('loss')[1].strip() for ln in o[0].splitlines() if 'loss' in ln] for q, o in outs.items()}
    assert loss['1'] == loss[None] and len(loss['1']) == 1


def test_an_in_kernel_wait_that_runs_out_is_redone_with_events(ctx):
    """The waits inside kernels that synchronise the two streams of the device loop are bounded (~1 - 3 s).  A second stream that is
    held up for longer - LCMI_REG_DELAY_US holds the regulariser chain back by five seconds per iteration here - makes the
    update's wait run out; the library notices at the end of the run, restores the state it copied at the start, switches the
    object to events and runs the same iterations again (lc_joint_run_adabelief): the numbers of the undisturbed fit, no error."""
    ds = make_roi_dataset(E=8, M=2, n=64, ss=2, seed=104)
    a = _fit(ctx, ds, 2, 2)
    b = _fit(ctx, ds, 2, 2, env={'LCMI_REG_DELAY_US': '5000000'})
    np.testing.assert_array_equal(a[0], b[0])
    for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'):
        np.testing.assert_array_equal(a[1][k], b[1][k])
