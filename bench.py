#!/usr/bin/env python3
"""Throughput of the PSF-fit hot loop (BASELINE.json metric: cutouts/sec), one rank per GPU.

A *step* is one launch of the persistent PSF-fit kernel: ITERS_PER_STEP AdaBelief iterations
(forward model + chi2 + full gradient + starlet l1 + fused update) over every stamp of the rank's
batch.  Workload at N = 1: BASELINE.json configs[1] (C2) = 100 frames x 8 stars, 32 x 32 stamps,
subsampling 2, starlet-regularised pixel-grid stage of the PSF fit; for N > 1 every rank gets its
own C2-sized batch (frames shard embarrassingly, no data-path collective => weak scaling).
Inputs are resident in HBM before the timed region; the Moffat stage and noise propagation are
one-time setup and not timed.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITERS_PER_STEP = 100
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_cutout_iteration(n, ss, S):
    """B_psf = 8 n^2 + (24 + 4 J) N^2 / S  (fp32)."""
    N = n * ss
    J = int(math.log2(N))
    return 8 * n * n + (24 + 4 * J) * N * N / S


def cpu_baseline(ds, ss, n_frames=3, n_iter=100, threads=8):
    """The float64 oracle (kind 'port') timed on the host cores on a bounded sample of the same
    workload: n_frames frames x S stamps x n_iter AdaBelief iterations of the pixel-grid stage.
    torch is limited to `threads` intra-op threads (more only oversubscribes these small FFTs)."""
    import torch
    from oracle import model as om, optim as oo
    from tests import helpers as H
    threads = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    S = ds['data'].shape[1]
    t_total = 0.0
    for f in range(n_frames):
        data, sig2, mask = H.psf_oracle_inputs(ds, f, ss)
        p = H.psf_initial_params(ds, f, ss)
        W = om.propagate_noise_psf(p, sig2, mask, ss)
        fn = lambda q: om.psf_loss(q, data, sig2, mask, ss, W=W, lam_scales=1.0, lam_hf=1.0)
        t0 = time.perf_counter()
        oo.adabelief(fn, p, ['B', 'a', 'x0', 'y0'], 1e-4, n_iter, schedule=True)
        t_total += time.perf_counter() - t0
    return dict(value=n_frames * S * n_iter / t_total, unit='cutouts/sec', cores=threads,
                kind='port',
                sample=f'{n_frames} frames x {S} stamps x {n_iter} AdaBelief iterations of the same C2 data, '
                       'torch float64 oracle')


def joint_fit_secondary(ctx, iters=100):
    """Secondary figure (not the contract metric): joint ROI forward-model iterations of BASELINE.json
    configs[3] (C4: 200 epochs, 64x64 ROI, 2 point sources + starlet-regularised background) on this GPU."""
    from lightcurver_amd.joint import JointFit
    from lightcurver_amd.synthetic import make_roi_dataset
    E, n, M, ss = 200, 64, 2, 2
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=104)
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, ctx)
    p = dict(ds['truth'])
    p['a'] = p['a'] * 0.9
    j.set_params(**p)
    W = j.propagate_noise()
    # the reference's code fall-backs of the main ROI optimisation (roi_modelling.py:308-312)
    j.set_loss(W=W, lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
    j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
    ab = dict(init_learning_rate=1e-4, schedule_learning_rate=False)
    j.run_adabelief(5, **ab)
    ctx.synchronize()
    ctx.timer_start()
    j.run_adabelief(iters, **ab)
    ms = ctx.timer_stop()
    hist = j.loss_history()
    N = n * ss
    J = int(math.log2(N))
    bytes_per = 8 * n * n + 4 * N * N + (24 + 4 * J) * N * N / E
    flops_per = 10.0 * (2 * N) ** 2 * math.log2((2 * N) ** 2) + 40.0 * N * N  # SURVEY.md 8(d), FFT route
    rate = E * iters / (ms * 1e-3)
    j.close()
    return {'workload': f'C4: {E} epochs x {n}x{n} ROI, {M} point sources + background, all parameters free, reference-default regularisation strengths, '
                        f'{iters} AdaBelief iterations (3 launches per iteration)',
            'cutouts_per_sec': rate, 'us_per_iteration': ms * 1e3 / iters,
            'algorithmic_bytes_per_cutout_iteration': bytes_per,
            'hbm_roofline_frac': rate * bytes_per / 1e9 / HBM_PEAK_GBS,
            'algorithmic_flop_per_cutout_iteration': flops_per, 'fp32_valu_frac_of_157': rate * flops_per / 1e12 / 157.3,
            'loss_finite': bool(np.all(np.isfinite(hist)))}


def c3_shard_secondary(ctx, iters=50):
    """Secondary figure: one GPU's share of BASELINE.json configs[2] (C3: 500 frames x 8 stars x 64x64 over 8
    GPUs = 63 frames per GPU), pixel-grid stage of the PSF fit."""
    from lightcurver_amd.psf_batch import PsfBatch
    from lightcurver_amd.synthetic import make_psf_dataset
    F, S, n, ss = 63, 8, 64, 2
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=103)
    weight = (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)
    b = PsfBatch(ds['data'], weight, ss, ctx)
    g = ds['fwhm_guess']
    f0 = np.sqrt(np.maximum(g * g - (2.0 / ss) ** 2, 1.0))
    b.set_moffat(np.stack([f0, f0, np.zeros(F), np.full(F, 2.5)], axis=-1))
    stars = np.zeros((F, S, 4), np.float32)
    stars[..., 0] = (ds['data'] * ds['masks']).sum(axis=(-1, -2))
    b.set_stars(stars)
    b.set_grid(None)
    b.fit_moffat(30)
    b.propagate_noise()
    b.set_regularization(None, 1.0, 1.0)
    ab = dict(init_learning_rate=1e-4, schedule_learning_rate=True)
    b.run_adabelief(5, **ab)
    ctx.synchronize()
    ctx.timer_start()
    b.run_adabelief(iters, **ab)
    ms = ctx.timer_stop()
    hist = b.loss_history()
    bytes_per = algorithmic_bytes_per_cutout_iteration(n, ss, S)
    rate = F * S * iters / (ms * 1e-3)
    b.close()
    return {'workload': f'C3 shard: {F} frames x {S} stars, {n}x{n} stamps (1/8 of C3), {iters} AdaBelief iterations',
            'cutouts_per_sec': rate, 'us_per_iteration': ms * 1e3 / iters,
            'algorithmic_bytes_per_cutout_iteration': bytes_per,
            'hbm_roofline_frac': rate * bytes_per / 1e9 / HBM_PEAK_GBS, 'loss_finite': bool(np.all(np.isfinite(hist)))}


def sharded_joint_fit(ctx, rank, world, iters=100):
    """Opt-in (--sharded-joint): C4's 200 epochs sharded over the ranks, shared block all-reduced in place by
    RCCL every iteration (lightcurver_amd/distributed.py).  Strong scaling: the total work is fixed."""
    import datetime
    import torch
    import torch.distributed as dist
    from lightcurver_amd.distributed import ShardedJointOptimizer, shard_epochs
    from lightcurver_amd.joint import JointFit
    from lightcurver_amd.synthetic import make_roi_dataset
    E, n, M, ss = 200, 64, 2, 2
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=104)  # same seed on every rank: identical full problem
    lo, hi = shard_epochs(E, world, rank)
    j = JointFit(ds['data'][lo:hi], ds['noisemap'][lo:hi].astype(np.float64) ** 2, ds['psf'][lo:hi], ss, M, ctx)
    p = dict(ds['truth'])
    p['a'] = (np.asarray(p['a']).reshape(E, M) * 0.9)[lo:hi].reshape(-1)
    for k in ('dx', 'dy', 'alpha', 'mean'):
        p[k] = np.asarray(p[k])[lo:hi]
    j.set_params(**p)
    j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0)
    j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
    torch.cuda.set_device(ctx.stream()[1])
    group = dist.new_group(backend='nccl', timeout=datetime.timedelta(seconds=120))
    opt = ShardedJointOptimizer(j, group)
    ab = dict(init_learning_rate=1e-4, schedule_learning_rate=False)
    opt.run(5, **ab)
    ctx.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    opt.run(iters, **ab)
    ctx.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    hist = j.loss_history()
    j.close()
    return {'workload': f'C4 sharded: {E} epochs x {n}x{n} ROI over {world} ranks, RCCL all-reduce of the shared block',
            'cutouts_per_sec': E * iters / dt, 'us_per_iteration': dt * 1e6 / iters, 'scaling': 'strong',
            'device_collective': bool(opt._dev), 'loss_finite': bool(np.all(np.isfinite(hist)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='C2', choices=['C1', 'C2', 'C3'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-joint', action='store_true', help='skip the secondary joint-fit figure')
    ap.add_argument('--sharded-joint', action='store_true',
                    help='also time the epoch-sharded joint fit with the in-place RCCL all-reduce (every rank takes part)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        # frames shard with no data-path collective: the process group only carries the barrier and
        # the max-over-ranks of the timing, so a host-side (gloo) group is sufficient and keeps the
        # timed region free of foreign GPU work.
        dist.init_process_group('gloo', rank=rank, world_size=world)

    from lightcurver_amd import _lib
    from lightcurver_amd.psf_batch import PsfBatch
    from lightcurver_amd.synthetic import CONFIGS, make_psf_dataset

    cfg = dict(CONFIGS[args.config])
    cfg.pop('kind')
    cfg['seed'] += 1000 * rank
    ds = make_psf_dataset(**cfg)
    F, S, n, ss = cfg['F'], cfg['S'], cfg['n'], cfg['ss']

    # rehearsal of the multi-rank path on a one-GPU box: LCMI_BENCH_DEVICE=0 puts every rank on that device
    ctx = _lib.Context(int(os.environ.get('LCMI_BENCH_DEVICE', local_rank)))
    weight = (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)
    b = PsfBatch(ds['data'], weight, ss, ctx)
    # setup (not timed): Moffat stage from the seeing guess, then noise propagation for the l1 weights
    g = ds['fwhm_guess']
    f0 = np.sqrt(np.maximum(g * g - (2.0 / ss) ** 2, 1.0))
    b.set_moffat(np.stack([f0, f0, np.zeros(F), np.full(F, 2.5)], axis=-1))
    stars = np.zeros((F, S, 4), np.float32)
    stars[..., 0] = (ds['data'] * ds['masks']).sum(axis=(-1, -2))
    b.set_stars(stars)
    b.set_grid(None)
    t0 = time.perf_counter()
    b.fit_moffat(100)
    b.propagate_noise()
    ctx.synchronize()
    setup_s = time.perf_counter() - t0
    b.set_regularization(None, 1.0, 1.0)  # strengths; W stays the propagated one on the device

    ab = dict(init_learning_rate=1e-4, schedule_learning_rate=True)
    for _ in range(args.warmup):
        b.run_adabelief(ITERS_PER_STEP, **ab)
    ctx.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        b.run_adabelief(ITERS_PER_STEP, **ab)
    kernel_ms = ctx.timer_stop()  # HIP events on the stream the kernels run on; also synchronises
    ctx.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        import torch
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    hist = b.loss_history()
    finite = bool(np.all(np.isfinite(hist)))
    res = b.results()
    sharded = None
    if args.sharded_joint:
        import torch.distributed as tdist
        own_group = False
        if not tdist.is_initialized():
            tdist.init_process_group('gloo', init_method='tcp://127.0.0.1:29531', rank=0, world_size=1)
            own_group = True
        try:
            sharded = sharded_joint_fit(ctx, rank, world)
        except Exception as e:
            sharded = {'error': repr(e)}
        if own_group:
            tdist.destroy_process_group()

    if rank == 0:
        cutouts = F * S * world
        value = cutouts * ITERS_PER_STEP * args.steps / elapsed
        bytes_per = algorithmic_bytes_per_cutout_iteration(n, ss, S)
        launch_s = kernel_ms * 1e-3 / args.steps
        achieved = F * S * ITERS_PER_STEP * bytes_per / launch_s / 1e9
        N = n * ss
        flops_per = 70.0 * N * N  # separable passes: 35 N^2 MAC per stamp-iteration
        traffic = None
        prof = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
        if os.path.exists(prof) and args.config == 'C2':
            try:
                traffic = json.load(open(prof)).get('psf_fit_kernel', {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        out = {
            'metric': 'cutouts/sec (PSF-fit + joint forward-model iter), 32x32 & 64x64 stamps',
            'value': value, 'unit': 'cutouts/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.config}: {F} frames x {S} stars, {n}x{n} stamps, subsampling {ss}, '
                                   f'starlet-regularised PSF pixel-grid fit, {ITERS_PER_STEP} AdaBelief '
                                   'iterations per step (one persistent launch)',
                       'frames_per_gpu': F, 'stars': S, 'stamp': n, 'subsampling': ss,
                       'iters_per_step': ITERS_PER_STEP, 'sharding': f'frames x{world} (no collective)',
                       'setup_seconds_untimed': setup_s, 'loss_finite': finite,
                       'median_reduced_chi2': float(np.median(res['chi2']))},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_source': 'profiles/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, bytes per launch)' if traffic else None,
                         'kernel': 'psf_fit_kernel', 'kernel_ms_per_launch': launch_s * 1e3,
                         'algorithmic_bytes_per_cutout_iteration': bytes_per,
                         'note': 'state is kept on-chip across iterations, so algorithmic bytes/s may exceed '
                                 'what HBM actually moves; fp32 VALU fraction reported beside it',
                         'fp32_valu_tflops': F * S * ITERS_PER_STEP * flops_per / launch_s / 1e12,
                         'fp32_valu_frac_of_157': F * S * ITERS_PER_STEP * flops_per / launch_s / 1e12 / 157.3},
        }
        try:  # the box's own copy bandwidth beside the 8 TB/s specification
            import ctypes
            g = ctypes.c_float()
            ctx.check(_lib.lib().lc_copy_bandwidth(ctx.h, 1 << 30, 10, ctypes.byref(g)), 'lc_copy_bandwidth')
            out['roofline']['measured_copy_GBps'] = g.value
            out['roofline']['frac_of_measured_copy'] = achieved / g.value
        except Exception as e:
            out['roofline']['measured_copy_GBps'] = None
        if world == 1 and not args.no_joint:
            try:
                out['config']['joint_fit'] = joint_fit_secondary(ctx)
            except Exception as e:
                out['config']['joint_fit'] = {'error': repr(e)}
            try:
                out['config']['c3_shard'] = c3_shard_secondary(ctx)
            except Exception as e:
                out['config']['c3_shard'] = {'error': repr(e)}
        if sharded is not None:
            out['config']['sharded_joint_fit'] = sharded
        if not args.no_cpu_baseline and world == 1:
            try:
                out['cpu_baseline'] = cpu_baseline(ds, ss)
            except Exception as e:  # the bench line must still be printed
                out['cpu_baseline'] = {'value': None, 'error': repr(e)}
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
