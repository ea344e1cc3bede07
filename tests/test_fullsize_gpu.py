"""Size-independent properties at BASELINE.json's full problem sizes, where the float64 oracle is too slow to be
the checker: linearity of the forward model in the fluxes, invariance under a permutation of the epochs,
frame independence of the PSF batch, equality of sharded and unsharded joint steps, monotone first iterations.
Tolerances: fp32 summation-order effects only (1e-5 relative) unless a bitwise statement is made."""
import numpy as np
import pytest

from lightcurver_amd.synthetic import make_psf_dataset, make_roi_dataset

pytestmark = pytest.mark.gpu


def _joint(ctx, ds, idx=None, M=2, ss=2):
    from lightcurver_amd.joint import JointFit
    sel = slice(None) if idx is None else idx
    j = JointFit(ds['data'][sel], ds['noisemap'][sel].astype(np.float64) ** 2, ds['psf'][sel], ss, M, ctx)
    p = {k: np.asarray(v, dtype=np.float64) for k, v in ds['truth'].items()}
    E = ds['data'].shape[0]
    if idx is not None:
        p['a'] = p['a'].reshape(E, M)[idx].reshape(-1)
        for k in ('dx', 'dy', 'alpha', 'mean'):
            p[k] = p[k][idx]
    return j, p


def test_c4_forward_is_linear_in_the_fluxes_and_permutation_invariant(ctx):
    """C4: 200 epochs x 64 x 64 ROI, 2 point sources + background."""
    E, M, n = 200, 2, 64
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
    j, p = _joint(ctx, ds)
    p0 = dict(p, h=np.zeros_like(p['h']), mean=np.zeros(E))
    j.set_params(**p0)
    m1, _ = j.model()
    j.set_params(**dict(p0, a=2.5 * p0['a']))
    m2, _ = j.model()
    assert np.abs(m2 - 2.5 * m1).max() <= 2e-6 * np.abs(m2).max()
    # superposition of background and point sources
    j.set_params(**dict(p, mean=np.zeros(E)))
    mfull, _ = j.model()
    j.set_params(**dict(p, a=np.zeros_like(p['a']), mean=np.zeros(E)))
    mbg, _ = j.model()
    assert np.abs(mfull - (m1 + mbg)).max() <= 3e-6 * np.abs(mfull).max()
    # epoch permutation: same loss, same shared gradients, per-epoch gradients permuted
    j.set_params(**p)
    j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0, lam_flux_uniformity=0.5)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean']
    j.set_free(free)
    loss, g = j.loss_grad(free)
    perm = np.random.default_rng(0).permutation(E)
    jp, pp = _joint(ctx, ds, perm)
    jp.set_params(**pp)
    jp.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0, lam_flux_uniformity=0.5)
    jp.set_free(free)
    lossp, gp = jp.loss_grad(free)
    assert abs(lossp - loss) <= 1e-5 * abs(loss)
    for k in ('c_x', 'c_y', 'h'):
        assert np.abs(gp[k] - g[k]).max() <= 2e-5 * np.abs(g[k]).max(), k
    assert np.abs(gp['a'].reshape(E, M) - g['a'].reshape(E, M)[perm]).max() <= 1e-5 * np.abs(g['a']).max()
    assert np.abs(gp['dx'] - g['dx'][perm]).max() <= 1e-5 * np.abs(g['dx']).max()


def test_c4_sharded_steps_equal_the_unsharded_fit(ctx):
    """Two shards of C4 stepped on one GPU with the shared block summed by hand = the single-object fit
    (the N = 2 arithmetic of lightcurver_amd/distributed.py at full size)."""
    E, M, n, T = 200, 2, 64, 5
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104)
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'h', 'mean']
    full, p = _joint(ctx, ds)
    full.set_params(**p)
    full.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0, lam_flux_uniformity=0.5)
    full.set_free(free)
    full.run_adabelief(T, init_learning_rate=1e-3)
    shards = []
    for idx in (np.arange(0, 100), np.arange(100, 200)):
        j, q = _joint(ctx, ds, idx)
        j.set_params(**q)
        j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0, lam_flux_uniformity=0.5)
        j.set_free(free)
        j.set_flux_reference(full.get_flux_reference())  # one reference for every shard (distributed.py does this)
        shards.append(j)
    for _ in range(T):
        bufs = []
        for j in shards:
            j.step_local()
            bufs.append(j.shared_get())
        tot = bufs[0] + bufs[1]
        for j in shards:
            j.shared_set(tot)
            j.step_update(init_learning_rate=1e-3)
    hf = full.loss_history()
    pf = full.get_params()
    ps = [j.get_params() for j in shards]
    assert np.abs(ps[0]['h'] - ps[1]['h']).max() == 0.0  # replicas stay in lock step, bit for bit
    assert np.abs(ps[0]['h'] - pf['h']).max() <= 2e-4 * T * 1e-3 + 1e-7
    assert np.abs(np.concatenate([ps[0]['a'], ps[1]['a']]) - pf['a']).max() <= 1e-5 * np.abs(pf['a']).max()
    # a shard's history holds the total loss of every iteration, evaluated from the summed block (its last entry
    # is recomputed from the local epochs only)
    hs = shards[0].loss_history()
    assert np.abs(hs[:T] - hf[:T]).max() <= 1e-5 * np.abs(hf).max()


def test_c2_frames_are_independent(ctx):
    """C2: 100 frames x 8 stars x 32 x 32.  A frame's fit does not depend on its neighbours in the batch: the first
    ten frames fitted alone give bit-identical grids, stars and loss histories."""
    from lightcurver_amd.psf_batch import PsfBatch
    F, S, n, ss, T = 100, 8, 32, 2, 40
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=102)
    w = (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)
    out = []
    for sub in (slice(0, F), slice(0, 10)):
        b = PsfBatch(ds['data'][sub], w[sub], ss, ctx)
        g = ds['fwhm_guess'][sub]
        f0 = np.sqrt(np.maximum(g * g - 1.0, 1.0))
        k = len(g)
        b.set_moffat(np.stack([f0, f0, np.zeros(k), np.full(k, 2.5)], axis=-1))
        st = np.zeros((k, S, 4), np.float32)
        st[..., 0] = (ds['data'][sub] * ds['masks'][sub]).sum(axis=(-1, -2))
        b.set_stars(st)
        b.set_grid(None)
        b.propagate_noise()
        b.set_regularization(None, 1.0, 1.0)
        b.run_adabelief(T, init_learning_rate=1e-4)
        out.append((b.loss_history(), b.get_grid(), b.get_stars()))
    for a, c in zip(out[0], out[1]):
        np.testing.assert_array_equal(a[:10], c)
    assert np.all(out[0][0][:, -1] < out[0][0][:, 0])


def test_c3_all_500_frames_on_one_gpu(ctx):
    """C3 at its full frame count on one GPU (500 frames x 8 stars x 64 x 64; the 8-GPU run shards it 63 frames per rank):
    four rounds of workgroups through the N = 128 kernel.  A frame's fit does not depend on the batch it is in: the last ten
    frames fitted alone give bit-identical grids, stars and loss histories; every frame's loss decreases."""
    from lightcurver_amd.psf_batch import PsfBatch
    F, S, n, ss, T = 500, 8, 64, 2, 40
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=103)
    w = (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)
    out = []
    for sub in (slice(0, F), slice(F - 10, F)):
        b = PsfBatch(ds['data'][sub], w[sub], ss, ctx)
        g = ds['fwhm_guess'][sub]
        f0 = np.sqrt(np.maximum(g * g - 1.0, 1.0))
        k = len(g)
        b.set_moffat(np.stack([f0, f0, np.zeros(k), np.full(k, 2.5)], axis=-1))
        st = np.zeros((k, S, 4), np.float32)
        st[..., 0] = (ds['data'][sub] * ds['masks'][sub]).sum(axis=(-1, -2))
        b.set_stars(st)
        b.set_grid(None)
        b.propagate_noise()
        b.set_regularization(None, 1.0, 1.0)
        b.run_adabelief(T, init_learning_rate=1e-4)
        out.append((b.loss_history(), b.get_grid(), b.get_stars()))
        b.close()
    for a, c in zip(out[0], out[1]):
        np.testing.assert_array_equal(a[F - 10:], c)
    assert np.all(np.isfinite(out[0][0])) and np.all(out[0][0][:, -1] < out[0][0][:, 0])


def test_c5_sized_epochs_decrease_the_loss(ctx):
    """C5 stamp size (128 x 128 ROI, 4 sources) on 16 epochs: finite, decreasing loss, fluxes move towards the truth."""
    E, M, n = 16, 4, 128
    ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=105)
    j, p = _joint(ctx, ds, M=M)
    start = dict(p, a=0.8 * p['a'], h=np.zeros_like(p['h']))
    j.set_params(**start)
    j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0)
    j.set_free(['a', 'dx', 'dy', 'mean', 'h'])
    j.run_adabelief(60, init_learning_rate=2e-3)
    h = j.loss_history()
    assert np.all(np.isfinite(h)) and h[-1] < 0.7 * h[0]
    got = j.get_params()['a']
    assert np.abs(got - p['a']).mean() < np.abs(start['a'] - p['a']).mean()
