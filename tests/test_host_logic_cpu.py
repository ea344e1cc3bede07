"""Host-side logic that needs no GPU: kwargs bookkeeping of the STARRED mirror, priors, initial guesses,
epoch sharding, and the sharded-optimiser protocol on two gloo ranks with the oracle as the local model."""
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

from oracle import model as om, optim as oo


def _kwargs(E=3, M=2, N=8):
    return {'kwargs_analytic': {'a': np.arange(E * M, dtype=float), 'c_x': np.array([1., 2.]), 'c_y': np.array([3., 4.]),
                                'dx': np.zeros(E), 'dy': np.ones(E), 'alpha': np.zeros(E)},
            'kwargs_background': {'h': np.zeros(N * N), 'mean': np.zeros(E)}, 'kwargs_sersic': {}}


def test_parameters_free_fixed_and_roundtrip():
    from lightcurver_amd.starred.deconvolution.parameters import ParametersDeconv
    k = _kwargs()
    fixed = deepcopy(k)
    del fixed['kwargs_analytic']['dx']
    del fixed['kwargs_analytic']['a']
    p = ParametersDeconv(k, fixed, deepcopy(k), deepcopy(k))
    assert p.free == ['a', 'dx'] and p.num_parameters == 9
    x = p.current_values()
    assert np.array_equal(x, np.concatenate([np.arange(6.), np.zeros(3)]))
    kw = p.args2kwargs(x + 1)
    assert np.allclose(kw['kwargs_analytic']['a'], np.arange(6.) + 1) and np.allclose(kw['kwargs_analytic']['dy'], 1)
    # the kwargs support the in-place manipulations the reference performs (roi_modelling.py:99-103)
    kw['kwargs_background']['h'] *= 0.0
    kw['kwargs_analytic']['a'] *= 0.0
    assert kw['kwargs_analytic']['a'].sum() == 0
    best = p.best_fit_values(as_kwargs=True)
    assert set(best) == {'kwargs_analytic', 'kwargs_background', 'kwargs_sersic'}
    with pytest.raises(NotImplementedError):
        f2 = deepcopy(k)
        del f2['kwargs_analytic']['alpha']
        ParametersDeconv(k, f2)


def test_prior_arrays():
    from lightcurver_amd.starred.deconvolution.loss import Prior
    pr = Prior(prior_analytic=[['c_x', np.array([1., 2.]), np.array([0.5, 0.5])]])
    arr = pr.as_arrays(2, [1., 2.], [3., 4.])
    assert np.allclose(arr['c_x_sigma'], 0.5) and np.all(arr['c_y_sigma'] > 1e12) and np.allclose(arr['c_y_mean'], [3, 4])
    with pytest.raises(NotImplementedError):
        Prior(prior_analytic=[['dx', 0., 1.]])


def test_initial_positions():
    from lightcurver_amd.starred.procedures.psf_routines import _initial_positions
    n = 16
    yy, xx = np.mgrid[0:n, 0:n]
    img = np.exp(-0.5 * ((xx - 9.0) ** 2 + (yy - 6.0) ** 2) / 2.0)[None]
    m = np.ones_like(img)
    assert _initial_positions(img, m, 'center') == (0.0, 0.0) or np.allclose(_initial_positions(img, m, 'center'), 0)
    x0, y0 = _initial_positions(img, m, 'barycenter')
    assert abs(x0[0] - 1.5) < 0.05 and abs(y0[0] + 1.5) < 0.05
    x0, y0 = _initial_positions(img, m, 'max')
    assert x0[0] == 1.5 and y0[0] == -1.5


def test_psf_stamp_preparation_follows_the_reference_rules():
    from lightcurver_amd.processes.psf_modelling import prepare_psf_stamps, relative_loss_differential
    rng = np.random.default_rng(0)
    d = rng.standard_normal((3, 10, 10))
    nm = np.ones((3, 10, 10))
    cosmics = np.zeros((3, 10, 10), bool)
    cosmics[1, :5, :] = True            # 50 % masked -> dropped
    cosmics[2, 0, :4] = True            # 4 % masked -> kept
    d[0, 3, 3] = np.nan
    nm[0, 3, 3] = np.nan                # both NaN -> (0, 1), masked
    d[0, 4, 4] = np.nan                 # only data NaN -> left alone (the C ABI zero-weights it later)
    dd, nn, mm, keep = prepare_psf_stamps(d, nm, cosmics)
    assert keep.tolist() == [True, False, True] and dd.shape == (2, 10, 10)
    assert dd[0, 3, 3] == 0.0 and nn[0, 3, 3] == 1.0 and not mm[0, 3, 3]
    assert np.isnan(dd[0, 4, 4]) and mm[0, 4, 4]
    assert mm[1].sum() == 96
    lh = np.concatenate([np.linspace(10, 1, 90), np.linspace(1, 0.9, 10)])
    assert abs(relative_loss_differential(lh) - 0.1 / 9.0) < 1e-12


def test_mask_surrounding_stars_keeps_the_central_object():
    from lightcurver_amd.processes.psf_modelling import mask_surrounding_stars
    n = 32
    yy, xx = np.mgrid[0:n, 0:n]
    star = lambda x0, y0, a: a * np.exp(-0.5 * ((xx - x0) ** 2 + (yy - y0) ** 2) / 2.0 ** 2)
    data = star(15.3, 15.8, 100.0) + star(5.0, 26.0, 60.0) + 0.1 * np.random.default_rng(0).standard_normal((n, n))
    noise = np.ones((n, n))
    m = mask_surrounding_stars(data, noise)
    assert m.dtype == bool and m.shape == (n, n)
    assert m[16, 15] and not m[26, 5]            # central star kept, neighbour masked
    assert (~m).sum() >= 15
    assert mask_surrounding_stars(0.1 * np.ones((n, n)), noise).all()  # nothing detected: all good


def test_star_epoch_preparation_downweights_whole_epochs():
    from lightcurver_amd.processes.star_photometry import prepare_star_epochs
    d = np.ones((4, 6, 6))
    nm = np.full((4, 6, 6), 2.0)
    cm = np.zeros((4, 6, 6), bool)
    cm[2, 1, 1] = cm[2, 2, 2] = True     # two flagged pixels, one epoch: x1000 once
    d[0, 0, 0] = nm[0, 0, 0] = np.nan
    dd, nn = prepare_star_epochs(d, nm, cm)
    assert dd[0, 0, 0] == 0.0 and nn[0, 0, 0] == 1e7
    assert np.all(nn[2] == 2000.0) and np.all(nn[1] == 2.0) and np.all(nn[3] == 2.0)


def test_sigma_clipped_stack_rejects_outliers():
    from lightcurver_amd.processes.roi_modelling import sigma_clipped_weighted_stack
    rng = np.random.default_rng(1)
    data = 5.0 + 0.1 * rng.standard_normal((30, 8, 8))
    data[3, 2, 2] = 500.0  # cosmic
    noise = np.full_like(data, 0.1)
    st = sigma_clipped_weighted_stack(data, noise)
    assert st.shape == (8, 8) and abs(st[2, 2] - 5.0) < 0.1 and abs(st.mean() - 5.0) < 0.05


def test_epoch_sharding_covers_everything_once():
    from lightcurver_amd.distributed import shard_epochs, shard_kwargs
    for E in (1, 7, 200, 1000):
        for world in (1, 2, 3, 8):
            blocks = [shard_epochs(E, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == E
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    flat = dict(a=np.arange(12.), c_x=np.zeros(2), c_y=np.zeros(2), dx=np.arange(6.), dy=np.arange(6.), alpha=np.zeros(6),
                h=np.zeros(4), mean=np.zeros(6))
    s = shard_kwargs(flat, 6, 2, 2, 1)
    assert np.array_equal(s['a'], np.arange(6., 12.)) and np.array_equal(s['dx'], [3., 4., 5.])


# ---- the sharded optimiser protocol, two gloo ranks, oracle as the local model -------------------------
class OracleLocalFit:
    """Same four methods as lightcurver_amd.joint.JointFit (step_local / shared_get / shared_set /
    step_update) with the float64 oracle doing the arithmetic of the local epochs."""

    def __init__(self, data, sig2, psf, ss, params, lam):
        self.data, self.sig2, self.psf, self.ss, self.lam = data, sig2, psf, ss, lam
        self.p = {k: v.clone() for k, v in params.items()}
        self.E, self.n = data.shape[0], data.shape[-1]
        self.M = self.p['c_x'].numel()
        self.N = ss * self.n
        self.free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean']
        self.m = {k: torch.zeros_like(self.p[k]) for k in self.free}
        self.s = {k: torch.zeros_like(self.p[k]) for k in self.free}
        self.t = 0
        self.losses = []

    def step_local(self):
        fn = lambda q: 0.5 * (((self.data - om.deconv_model(q, self.psf, self.ss, self.n)) ** 2) / self.sig2).sum()
        L, g = oo.value_and_grad(fn, self.p, self.free)
        self.g = g
        a2 = self.p['a'].reshape(self.E, self.M)
        self.shared = torch.cat([g['h'], g['c_x'], g['c_y'], a2.sum(0), (a2 ** 2).sum(0),
                                 torch.tensor([2 * L, float(self.E)], dtype=om.DT)])

    def shared_get(self):
        return self.shared.numpy().copy()

    def shared_set(self, buf):
        self.shared = torch.as_tensor(buf, dtype=om.DT)

    def step_update(self, init_learning_rate=1e-3, **_):
        NN, M = self.N * self.N, self.M
        sh = self.shared
        Etot = float(sh[NN + 4 * M + 1])
        g = dict(self.g)
        g['h'], g['c_x'], g['c_y'] = sh[:NN].clone(), sh[NN:NN + M], sh[NN + M:NN + 2 * M]
        h = self.p['h'].detach().requires_grad_(True)
        W = om.default_W(self.N, om.n_scales(self.N))
        reg = om.l1_starlet(h.reshape(self.N, self.N), W, self.lam['sc'], self.lam['hf'], om.n_scales(self.N)) \
            + self.lam['pos'] * torch.clamp(-h, min=0).sum()
        (gr,) = torch.autograd.grad(reg, h)
        g['h'] = g['h'] + gr
        mean = sh[NN + 2 * M:NN + 3 * M] / Etot
        var = torch.clamp(sh[NN + 3 * M:NN + 4 * M] / Etot - mean ** 2, min=0)
        sd = torch.sqrt(var)
        a2 = self.p['a'].reshape(self.E, M)
        g['a'] = g['a'] + (self.lam['fu'] * (a2 - mean) / (Etot * sd)).reshape(-1)
        self.losses.append(0.5 * float(sh[NN + 4 * M]) + float(reg) + self.lam['fu'] * float(sd.sum()))
        b1, b2, eps, er, t = 0.9, 0.999, 1e-16, 1e-16, self.t
        for k in self.free:
            self.m[k] = b1 * self.m[k] + (1 - b1) * g[k]
            self.s[k] = b2 * self.s[k] + (1 - b2) * (g[k] - self.m[k]) ** 2 + er
            self.p[k] = self.p[k].detach() - init_learning_rate * (self.m[k] / (1 - b1 ** (t + 1))) / (
                torch.sqrt(self.s[k] / (1 - b2 ** (t + 1))) + eps)
        self.t += 1


    # -- what lightcurver_amd.distributed.sharded_lbfgs drives (the L-BFGS-B stage): parameters in and out, and the
    #    gradient-only counterpart of step_update behind the all-reduce
    def set_free(self, names):
        self.lbfgs_free = list(names)

    def get_params(self):
        return {k: v.detach().numpy().copy() for k, v in self.p.items()}

    def set_params(self, **kw):
        for k, v in kw.items():
            self.p[k] = om.T(np.asarray(v, np.float64).reshape(self.p[k].shape))

    def step_grad(self, names):
        NN, M = self.N * self.N, self.M
        sh = self.shared
        Etot = float(sh[NN + 4 * M + 1])
        g = dict(self.g)
        g['c_x'], g['c_y'] = sh[NN:NN + M], sh[NN + M:NN + 2 * M]
        mean = sh[NN + 2 * M:NN + 3 * M] / Etot
        sd = torch.sqrt(torch.clamp(sh[NN + 3 * M:NN + 4 * M] / Etot - mean ** 2, min=0))
        a2 = self.p['a'].reshape(self.E, M)
        g['a'] = g['a'] + (self.lam['fu'] * (a2 - mean) / (Etot * sd)).reshape(-1)
        loss = 0.5 * float(sh[NN + 4 * M]) + self.lam['fu'] * float(sd.sum())
        return loss, {k: g[k].detach().numpy() for k in names}


def _problem():
    from lightcurver_amd.synthetic import make_roi_dataset
    ds = make_roi_dataset(E=4, M=2, n=8, ss=2, seed=17)
    rng = np.random.default_rng(3)
    p = {k: om.T(v) for k, v in ds['truth'].items()}
    p['a'] = p['a'] * om.T(rng.uniform(0.9, 1.1, p['a'].shape))
    p['h'] = p['h'] + om.T(1e-3 * rng.standard_normal(p['h'].shape))
    return ds, p, om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])


def _rank_main(rank, world, port, ret):
    import torch.distributed as dist
    from lightcurver_amd.distributed import ShardedJointOptimizer, gather_epoch_blocks, shard_epochs, shard_kwargs
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    ds, p, data, sig2, psf = _problem()
    lo, hi = shard_epochs(4, world, rank)
    flat = {k: v.numpy() for k, v in p.items()}
    loc = {k: om.T(v) for k, v in shard_kwargs(flat, 4, 2, world, rank).items()}
    fit = OracleLocalFit(data[lo:hi], sig2[lo:hi], psf[lo:hi], 2, loc, dict(sc=1.0, hf=1.0, pos=5.0, fu=0.4))
    ShardedJointOptimizer(fit).run(5, init_learning_rate=1e-3)
    full = gather_epoch_blocks({k: v.numpy() for k, v in fit.p.items()}, 2)
    if rank == 0:
        ret['params'] = full
        ret['losses'] = fit.losses
    dist.destroy_process_group()


def test_sharded_optimiser_equals_single_rank_gloo():
    import torch.multiprocessing as mp
    ds, p, data, sig2, psf = _problem()
    W = om.default_W(16, 4)
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, 2, W=W, lam_scales=1.0, lam_hf=1.0, lam_pos=5.0, lam_fu=0.4)
    pf, lh, l0 = oo.adabelief(fn, p, ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean'], 1e-3, 5, schedule=False)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_rank_main, args=(2, port, ret), nprocs=2, join=True)
    got = ret['params']
    assert np.allclose(ret['losses'], [l0] + lh[:-1], rtol=1e-9)
    for k in ('a', 'c_x', 'c_y', 'dx', 'dy', 'mean'):
        assert np.allclose(got[k], pf[k].numpy(), rtol=1e-7, atol=1e-9), k
    dh = np.abs(got['h'] - pf['h'].numpy())
    assert np.median(dh) < 1e-10 and dh.max() < 2.5e-3  # a sign flip of a ~0 gradient moves one pixel by <= 2 lr


def _rank_main_lbfgs(rank, world, port, ret):
    import torch.distributed as dist
    from lightcurver_amd.distributed import ShardedJointOptimizer, gather_epoch_blocks, shard_epochs, shard_kwargs
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    ds, p, data, sig2, psf = _problem()
    lo, hi = shard_epochs(4, world, rank)
    flat = {k: v.numpy() for k, v in p.items()}
    loc = {k: om.T(v) for k, v in shard_kwargs(flat, 4, 2, world, rank).items()}
    fit = OracleLocalFit(data[lo:hi], sig2[lo:hi], psf[lo:hi], 2, loc, dict(sc=0.0, hf=0.0, pos=0.0, fu=0.4))
    hist, res = ShardedJointOptimizer(fit).run_lbfgs(['a', 'c_x', 'c_y', 'dx', 'dy'], 30, lower={'a': 0.0})
    full = gather_epoch_blocks({k: v.numpy() for k, v in fit.p.items()}, 2)
    if rank == 0:
        ret['params'] = full
        ret['fun'] = float(res.fun)
        ret['hist'] = hist
    dist.destroy_process_group()


def test_sharded_lbfgs_stage_equals_single_rank_gloo():
    """lightcurver_amd.distributed.sharded_lbfgs on two gloo ranks (the oracle as the local model): the optimum of scipy's
    L-BFGS-B on the loss of all epochs in one piece."""
    import torch.multiprocessing as mp
    from scipy.optimize import minimize
    ds, p, data, sig2, psf = _problem()
    free = ['c_x', 'c_y', 'a', 'dx', 'dy']
    sizes = [p[k].numel() for k in free]
    offs = np.concatenate([[0], np.cumsum(sizes)])

    def fun(x):
        q = dict(p)
        for i, k in enumerate(free):
            q[k] = om.T(x[offs[i]:offs[i + 1]])
        L, g = oo.value_and_grad(lambda r: om.deconv_loss(r, data, sig2, psf, 2, lam_fu=0.4), q, free)
        return float(L), np.concatenate([g[k].numpy() for k in free])

    x0 = np.concatenate([p[k].numpy().ravel() for k in free])
    lo = np.concatenate([np.full(sz, 0.0 if k == 'a' else -np.inf) for k, sz in zip(free, sizes)])
    ref = minimize(fun, x0, jac=True, method='L-BFGS-B', bounds=list(zip(lo, np.full(lo.size, np.inf))), options=dict(maxiter=30))
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + os.getpid() % 2000
    mp.spawn(_rank_main_lbfgs, args=(2, port, ret), nprocs=2, join=True)
    assert abs(ret['fun'] - ref.fun) <= 1e-8 * abs(ref.fun)
    assert len(ret['hist']) >= 1 and ret['hist'][-1] <= ret['hist'][0]
    got = ret['params']
    for i, k in enumerate(free):
        assert np.allclose(got[k], ref.x[offs[i]:offs[i + 1]], rtol=1e-5, atol=1e-7), k


class _FailingPeer:
    """Stands in for distributed.PeerGroup on CPU: the callback is never called (the stand-in fit's run_sharded ignores it),
    check() reports a wait that ran out on ``bad_rank`` only - what csrc/peer.hip leaves behind after a time-out."""

    def __init__(self, rank, bad_rank):
        self.rank, self.bad = rank, bad_rank

    def callback(self):
        return None, None

    def check(self):
        if self.rank == self.bad:
            raise RuntimeError('peer all-reduce: a rank did not publish its block in time')


def _rank_main_errors(rank, world, port, ret):
    import torch.distributed as dist
    from lightcurver_amd.distributed import ShardedJointOptimizer, shard_epochs, shard_kwargs
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    ds, p, data, sig2, psf = _problem()
    lo, hi = shard_epochs(4, world, rank)
    flat = {k: v.numpy() for k, v in p.items()}
    loc = {k: om.T(v) for k, v in shard_kwargs(flat, 4, 2, world, rank).items()}
    fit = OracleLocalFit(data[lo:hi], sig2[lo:hi], psf[lo:hi], 2, loc, dict(sc=0.0, hf=0.0, pos=0.0, fu=0.4))
    out = {}
    # (1) AdaBelief loop over the peer transport: rank 1's wait ran out -> both ranks raise, naming rank 1
    fit.run_sharded = lambda n_iter, fn, user, **cfg: None
    opt = ShardedJointOptimizer(fit, peer=_FailingPeer(rank, 1))
    try:
        opt.run(3, init_learning_rate=1e-3)
        out['run'] = 'no error'
    except RuntimeError as e:
        out['run'] = str(e)
    del fit.run_sharded
    # (2) L-BFGS-B stage: rank 1's third evaluation fails -> both ranks raise at that evaluation, nobody hangs in a gather
    calls = [0]
    real = fit.step_local

    def flaky():
        calls[0] += 1
        if rank == 1 and calls[0] == 3:
            raise RuntimeError('device step failed (test)')
        real()

    fit.step_local = flaky
    try:
        ShardedJointOptimizer(fit).run_lbfgs(['a', 'dx', 'dy'], 30, lower={'a': 0.0})
        out['lbfgs'] = 'no error'
    except RuntimeError as e:
        out['lbfgs'] = str(e)
    out['calls'] = calls[0]
    ret[rank] = out
    dist.barrier()        # both ranks are still in step: the next collective finds its partner
    dist.destroy_process_group()


def test_a_failure_on_one_rank_raises_on_every_rank_gloo():
    """a peer wait that ran out (or any failing step) on ONE rank must stop ALL ranks together - the others would
    otherwise optimise on and hang in their next host collective until the gloo time-out."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + os.getpid() % 2000
    mp.spawn(_rank_main_errors, args=(2, port, ret), nprocs=2, join=True)
    for rank in (0, 1):
        assert 'rank 1' in ret[rank]['run'] and 'did not publish' in ret[rank]['run'], ret[rank]
        assert 'rank 1' in ret[rank]['lbfgs'] and 'device step failed' in ret[rank]['lbfgs'], ret[rank]
        assert ret[rank]['calls'] == 3


def test_blended_neighbour_is_split_off_and_masked():
    """A neighbour whose wings touch the central star forms ONE connected group with it above 3 sigma; sep's
    multi-threshold de-blending (deblend_cont = 0.001, psf_modelling.py:51-52) splits the group, so the neighbour is
    masked while the central star is kept.  Also: a faint bump below the contrast criterion is not split off."""
    from lightcurver_amd.processes.psf_modelling import mask_surrounding_stars
    from lightcurver_amd.processes.source_masking import extract
    n = 32
    yy, xx = np.mgrid[0:n, 0:n]

    def star(x0, y0, amp, s=1.8):
        return amp * np.exp(-0.5 * ((xx - x0) ** 2 + (yy - y0) ** 2) / s ** 2)

    rng = np.random.default_rng(5)
    noise = np.ones((n, n))
    data = star(15.5, 15.5, 400.0) + star(22.5, 17.0, 150.0) + rng.normal(0, 1.0, (n, n))
    objects, seg = extract(data, noise)
    assert len(objects) == 2                                       # one group above threshold, two objects after de-blending
    lab_c, lab_n = seg[15, 15], seg[17, 22]
    assert lab_c > 0 and lab_n > 0 and lab_c != lab_n
    m = mask_surrounding_stars(data, noise)
    assert m[15, 15] and m[13:18, 13:18].all()                      # central star kept
    assert not m[17, 22] and not m[16:19, 21:24].any()              # neighbour masked
    assert 15 <= (~m).sum() <= 120
    # contrast criterion: a bump carrying less than 0.1 % of the flux stays part of the central object
    data2 = star(15.5, 15.5, 4000.0, s=2.5) + star(23.0, 15.5, 3.0, s=1.0) + rng.normal(0, 1.0, (n, n))
    objects2, seg2 = extract(data2, noise)
    assert len(objects2) == 1 and mask_surrounding_stars(data2, noise).all()


def test_cleaning_merges_detections_that_exist_only_on_a_neighbours_wing():
    """sep's clean pass (clean=True, clean_param=1.0 are the defaults the reference's call leaves in place,
    psf_modelling.py:51-52): a fragment the de-blending split off that would not reach minarea pixels above the
    threshold on its own is merged into the neighbour whose Moffat wing lies under it; real neighbours stay."""
    from lightcurver_amd.processes.source_masking import extract, clean, _shape, matched_filter_snr
    n = 48
    yy, xx = np.mgrid[0:n, 0:n]

    def star(x0, y0, amp, s):
        return amp * np.exp(-0.5 * ((xx - x0) ** 2 + (yy - y0) ** 2) / s ** 2)

    noise = np.ones((n, n))
    data = star(23.5, 23.5, 3000.0, 2.0) + star(40, 23.5, 3.0, 2.5) + np.random.default_rng(7).normal(0, 1.0, (n, n))
    raw, _ = extract(data, noise, clean_detections=False)
    cleaned, seg = extract(data, noise)
    assert len(raw) == 3 and raw['npix'].min() < 15           # the de-blending leaves a fragment below minarea
    assert len(cleaned) == 2 and cleaned['npix'].min() >= 15
    assert cleaned['npix'].sum() == raw['npix'].sum()         # merged, not dropped: its pixels stay in the map
    assert sorted(np.unique(seg)) == [0, 1, 2]
    # the wing model: a Moffat of index 1 through the threshold at the isophotal area of the bright star
    snr = matched_filter_snr(data, noise)
    bright = seg == seg[23, 23]
    q = _shape(bright, snr, 3.0, 15)
    r_t2 = q['npix'] / q['unitarea']
    alpha = (q['amp'] / 3.0 - 1.0) * q['unitarea'] / q['npix']
    assert abs(q['amp'] / (1.0 + alpha * r_t2) - 3.0) < 1e-9
    # two real stars: nothing to clean
    two = star(15.5, 23.5, 400.0, 1.8) + star(30.5, 25.0, 150.0, 1.8) + np.random.default_rng(5).normal(0, 1.0, (n, n))
    assert len(extract(two, noise)[0]) == len(extract(two, noise, clean_detections=False)[0]) == 2
    assert clean([bright], snr, 3.0, 15) == [bright] or len(clean([bright], snr, 3.0, 15)) == 1
