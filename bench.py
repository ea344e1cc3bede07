#!/usr/bin/env python3
"""Throughput of the PSF-fit hot loop (BASELINE.json metric: cutouts/sec), one rank per GPU.

A *step* is one pass of the hot path over one batch: the whole pixel-grid stage of the PSF fit of the batch
(reference default ``psf_n_iter_pixels: 3000`` AdaBelief iterations, config.yaml:227; each iteration = forward model
+ chi2 + full gradient + starlet l1 + fused update of every stamp), from B = 0 at the Moffat-stage optimum, as ONE
persistent launch.  Workload at N = 1: BASELINE.json configs[1] (C2) = 100 frames x 8 stars, 32 x 32 stamps,
subsampling 2; for N > 1 every rank gets its own C2-sized batch (frames shard embarrassingly, no data-path
collective => weak scaling) and, beside it, C4's 200 epochs sharded over the ranks with the in-place RCCL all-reduce
of the shared block (strong scaling, reported under config.sharded_joint_fit).
Inputs are resident in HBM before the timed region; the Moffat stage and noise propagation are one-time setup.

`--gpus N` without a launcher (WORLD_SIZE unset) starts N rank processes itself before anything touches the GPU.
For N > 1 every rank process is a supervisor that stays off the GPU and runs the measurement in a child (supervise_rank):
the headline line exists before the sharded joint fits start and survives whatever happens in them.
"""
import argparse
import csv
import glob
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITERS_PER_STEP = 3000   # psf_n_iter_pixels (reference config.yaml:227)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3


def psf_bytes_per_cutout_iteration(n, ss, S):
    """B_psf = 8 n^2 + (24 + 4 J) N^2 / S  (fp32)."""
    N = n * ss
    J = int(math.log2(N))
    return 8 * n * n + (24 + 4 * J) * N * N / S


def joint_bytes_per_cutout_iteration(n, ss, E_local):
    """B_roi = 8 n^2 + 4 N^2 + (24 + 4 J) N^2 / E_local."""
    N = n * ss
    J = int(math.log2(N))
    return 8 * n * n + 4 * N * N + (24 + 4 * J) * N * N / E_local


def joint_flops_per_cutout_iteration(n, ss):
    N = n * ss
    return 10.0 * (2 * N) ** 2 * math.log2((2 * N) ** 2) + 40.0 * N * N  # SURVEY.md 8(d), FFT route


def joint_flops_executed_per_cutout_iteration(n, ss):
    """What the epoch kernel executes (csrc/joint_kernels.h) instead of SURVEY's (2 N)^2 transforms: length L = 3 N / 2 (the
    smallest alias-free 'same' window), two real rows per complex transform, half spectra, binned rows - per epoch N / 2
    complex row transforms of length L in phase A and N / 2 in phase C', L / 2 forward + L / 2 inverse column transforms in
    each of B and B', and n / 2 + n / 2 transforms of length L / 2 in phase C; 5 L log2 L flop per complex transform, plus
    ~40 N^2 for the spectrum products, scene, residual and gradient arithmetic."""
    N = n * ss
    L = 3 * N // 2
    return 5.0 * ((N + 2 * L) * L * math.log2(L) + n * (L / 2) * math.log2(L / 2)) + 40.0 * N * N


def hbm_roofline(bytes_per_launch, launch_s, kernel, extra=None):
    achieved = bytes_per_launch / launch_s / 1e9
    r = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
         'traffic': None, 'kernel': kernel, 'kernel_ms_per_launch': launch_s * 1e3,
         'algorithmic_bytes_per_launch': bytes_per_launch}
    if extra:
        r.update(extra)
    return r


# ---------------------------------------------------------------------------------------------------------------
# multi-rank launch without a launcher
# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """Start n rank processes of this script (fresh interpreters: nothing in this process has touched the GPU)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    return rc


def supervise_rank(argv, grace_s=600.0, script=None):
    """N > 1: the rank process the launcher started stays off the GPU and runs the bench as a child whose standard output
    it reads.  The child writes a provisional line (rank 0) and the marker 'HEADLINE_DONE' once the headline measurement
    is complete, then the final line after the sharded joint fits.  Whatever happens to the child in that last part - an
    RCCL or peer-memory fault ends a process without a Python exception, a hung collective never returns - the LAST
    complete line reaches this rank's standard output, once, and the launcher sees exit code 0: the headline number
    does not depend on the part of the bench that only a multi-GPU node can run.  After the marker the child has
    `grace_s` seconds; it is then killed by its exact PID."""
    import signal
    import threading
    env = dict(os.environ, LCMI_BENCH_WORKER='1')
    child = subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE)
    state = {'line': None, 'marker_at': None}

    def reader():
        for raw in child.stdout:
            text = raw.decode(errors='replace').strip()
            if text == 'HEADLINE_DONE':
                state['marker_at'] = time.monotonic()
            elif text.startswith('{'):
                state['line'] = text

    def forward(signum, frame):
        if child.poll() is None:
            child.kill()
        sys.exit(128 + signum)

    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, forward)
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    killed = False
    while True:
        try:
            child.wait(timeout=1.0)
            break
        except subprocess.TimeoutExpired:
            if state['marker_at'] is not None and time.monotonic() - state['marker_at'] > grace_s:
                print(f'bench.py: the sharded joint fits did not finish within {grace_s:.0f} s of the headline '
                      f'measurement; rank {os.environ.get("RANK")} ends its worker (pid {child.pid})', file=sys.stderr)
                child.kill()
                child.wait()
                killed = True
                break
    th.join(timeout=10.0)
    if state['line'] is not None:
        line = state['line']
        if killed or child.returncode != 0:
            print(f'bench.py: rank {os.environ.get("RANK")}: the worker ended before the sharded joint fits did (killed after the '
                  f'grace period: {killed}, exit code {child.returncode}); the headline line is kept', file=sys.stderr)
            try:
                d = json.loads(line)
                d['config'].setdefault('sharded_joint_fit', {})
                if 'value' not in d['config']['sharded_joint_fit']:
                    d['config']['sharded_joint_fit'] = {'error': 'the worker ended before the sharded joint fits did '
                                                                 f'(killed after the grace period: {killed}, exit code '
                                                                 f'{child.returncode})'}
                line = json.dumps(d)
            except Exception:
                pass
        sys.stdout.write(line + '\n')
        sys.stdout.flush()
    if state['marker_at'] is not None:
        return 0    # (the headline was measured: what happened afterwards is in the line's error field and on stderr)
    print(f'bench.py: rank {os.environ.get("RANK")}: the worker ended before the headline measurement (exit code '
          f'{child.returncode})', file=sys.stderr)
    return child.returncode if child.returncode is not None else 1


# ---------------------------------------------------------------------------------------------------------------
# HBM traffic of the timed kernel from the PMC counters (separate rocprofv3 passes, run before this process
# initialises the GPU; MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE in their own passes, KiB units,
# FETCH_SIZE doubled on gfx950, WRITE_SIZE exact for 16-byte stores)
# ---------------------------------------------------------------------------------------------------------------
def measure_traffic(timeout_s=150):
    exe = shutil.which('rocprofv3')
    if not exe:
        return None, 'rocprofv3 not found'
    out = {}
    tmp = tempfile.mkdtemp(prefix='lcmi_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, counter)
            cmd = [exe, '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable,
                   os.path.abspath(__file__), '--pmc-child']
            try:
                subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=timeout_s, check=True)
            except Exception as e:
                return None, f'{counter} pass failed: {e!r}'
            vals = []
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get('Counter_Name') != counter or 'psf_fit_kernel' not in r.get('Kernel_Name', ''):
                        continue
                    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
                    if us > 5000.0:   # the ITERS_PER_STEP launches, not the one-iteration evaluations of the setup
                        vals.append(float(r['Counter_Value']))
            if not vals:
                return None, f'no {counter} rows for the timed kernel'
            out[counter] = sum(vals) / len(vals)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    traffic = (2.0 * out['FETCH_SIZE'] + out['WRITE_SIZE']) * 1024.0
    return traffic, (f'rocprofv3 --pmc passes inside this run: FETCH_SIZE {out["FETCH_SIZE"]:.0f} KiB (x2 on gfx950) + '
                     f'WRITE_SIZE {out["WRITE_SIZE"]:.0f} KiB per {ITERS_PER_STEP}-iteration launch')


PMC_ITERS = 20   # iterations inside a marked section of the --pmc-child-sections run
# the sections of that run: key -> (what is iterated, rows of the section are divided by this many units)
PMC_SECTIONS = ('C4', 'C4 shard', 'C5 shard', 'C3 shard', 'star photometry')


def measure_section_traffic(timeout_s=240):
    """HBM traffic of ONE iteration of every secondary workload from the same two rocprofv3 --pmc passes as measure_traffic,
    on ONE child per counter that runs the workloads one after the other.  The measured iterations of a workload are bracketed
    explicitly: the child launches a marker dispatch (lc_ctx_marker, kernel lc_marker_kernel) before and after them, and only
    the rows between the two markers are summed - every kernel of the iteration (epoch kernel or phases, reduction + update,
    regulariser chain), nothing of the set-up, whatever the launch counts are.  LCMI_EVENT_SYNC=1: counter collection
    serialises the two streams, the update must wait for the chain by an event.
    Returns {key: (bytes per iteration or None, source text)}."""
    exe = shutil.which('rocprofv3')
    if not exe:
        return {k: (None, 'rocprofv3 not found') for k in PMC_SECTIONS}
    tot = {}
    tmp = tempfile.mkdtemp(prefix='lcmi_pmcs_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp', LCMI_EVENT_SYNC='1')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, counter)
            cmd = [exe, '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable,
                   os.path.abspath(__file__), '--pmc-child-sections']
            try:
                subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=timeout_s, check=True)
            except Exception as e:
                return {k: (None, f'{counter} pass failed: {e!r}') for k in PMC_SECTIONS}
            rows = []
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get('Counter_Name') == counter:
                        rows.append((int(r['Dispatch_Id']), r.get('Kernel_Name', ''), float(r['Counter_Value'])))
            rows.sort()
            sums, n_markers, open_section = {}, 0, None
            for _, name, val in rows:
                if 'lc_marker_kernel' in name:
                    n_markers += 1
                    open_section = (n_markers - 1) // 2 if n_markers % 2 == 1 else None
                elif open_section is not None and open_section < len(PMC_SECTIONS):
                    sums[open_section] = sums.get(open_section, 0.0) + val
            if n_markers != 2 * len(PMC_SECTIONS):
                return {k: (None, f'{counter}: {n_markers} markers in the counter rows, expected {2 * len(PMC_SECTIONS)}') for k in PMC_SECTIONS}
            tot[counter] = sums
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for i, key in enumerate(PMC_SECTIONS):
        fs, ws = tot['FETCH_SIZE'].get(i, 0.0) / PMC_ITERS, tot['WRITE_SIZE'].get(i, 0.0) / PMC_ITERS
        out[key] = ((2.0 * fs + ws) * 1024.0,
                    f'rocprofv3 --pmc passes inside this run, every kernel between the markers around {PMC_ITERS} iterations: '
                    f'FETCH_SIZE {fs:.0f} KiB (x2 on gfx950) + WRITE_SIZE {ws:.0f} KiB per iteration')
    return out


def pmc_sections_child():
    """under rocprofv3 --pmc: the secondary workloads, PMC_ITERS iterations of each between two marker dispatches; nothing timed"""
    from lightcurver_amd import _lib
    from lightcurver_amd.joint import JointFit, StarPhotometryBatch
    from lightcurver_amd.psf_batch import PsfBatch
    from lightcurver_amd.synthetic import make_psf_dataset, make_roi_dataset
    ctx = _lib.Context(0)
    tag = [0]

    def marked(fn):
        tag[0] += 1
        ctx.marker(tag[0])
        fn()
        ctx.synchronize()
        tag[0] += 1
        ctx.marker(tag[0])

    for E, n, M in ((200, 64, 2), (25, 64, 2), (125, 128, 4)):
        ds = make_roi_dataset(E=E, M=M, n=n, ss=2, seed=104 if n == 64 else 105)
        j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], 2, M, ctx)
        p = dict(ds['truth'])
        p['a'] = p['a'] * 0.9
        j.set_params(**p)
        W = j.propagate_noise()
        j.set_loss(W=W, lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
        j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
        ab = dict(init_learning_rate=1e-4, schedule_learning_rate=False)
        j.run_adabelief(5, **ab)
        marked(lambda: j.run_adabelief(PMC_ITERS, **ab))
        j.close()
    # C3 shard: the iterations of the pixel-grid stage are inside one launch
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
    marked(lambda: b.run_adabelief(PMC_ITERS, **ab))
    b.close()
    # star photometry batch
    G, E, n = 30, 100, 32
    base = make_roi_dataset(E=E, M=1, n=n, ss=2, seed=106, with_background=False)
    sig2 = base['noisemap'].astype(np.float64) ** 2
    sb = StarPhotometryBatch([(base['data'], sig2, base['psf'])] * G, 2, 1, ctx)
    a0 = np.asarray(base['truth']['a'], np.float64) * 0.9
    sb.set_params(a=np.tile(a0, G), c_x=np.zeros(G), c_y=np.zeros(G), dx=np.zeros(G * E), dy=np.zeros(G * E),
                  alpha=np.zeros(G * E), mean=np.zeros(G * E))
    sb.set_loss()
    sb.set_free(['a', 'c_x', 'c_y', 'dx', 'dy'])
    ab = dict(init_learning_rate=1e-3, schedule_learning_rate=True)
    sb.run_adabelief(5, **ab)
    marked(lambda: sb.run_adabelief(PMC_ITERS, **ab))
    sb.close()
    ctx.synchronize()


# ---------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------
def setup_psf_batch(ctx, cfg_name, rank=0, moffat_iters=100):
    from lightcurver_amd.psf_batch import PsfBatch
    from lightcurver_amd.synthetic import CONFIGS, make_psf_dataset
    cfg = dict(CONFIGS[cfg_name])
    cfg.pop('kind')
    cfg['seed'] += 1000 * rank
    ds = make_psf_dataset(**cfg)
    F, S, n, ss = cfg['F'], cfg['S'], cfg['n'], cfg['ss']
    weight = (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)
    b = PsfBatch(ds['data'], weight, ss, ctx)
    g = ds['fwhm_guess']
    f0 = np.sqrt(np.maximum(g * g - (2.0 / ss) ** 2, 1.0))
    b.set_moffat(np.stack([f0, f0, np.zeros(F), np.full(F, 2.5)], axis=-1))
    stars = np.zeros((F, S, 4), np.float32)
    stars[..., 0] = (ds['data'] * ds['masks']).sum(axis=(-1, -2))
    b.set_stars(stars)
    b.set_grid(None)
    t0 = time.perf_counter()
    b.fit_moffat(moffat_iters)
    b.propagate_noise()
    ctx.synchronize()
    setup_s = time.perf_counter() - t0
    b.set_regularization(None, 1.0, 1.0)  # strengths; W stays the propagated one on the device
    return ds, weight, b, b.get_stars(), setup_s, (F, S, n, ss)


def psf_step(b, stars0, ab):
    """One step: the pixel-grid stage from its start (B = 0, zero moments, stars at the Moffat-stage optimum)."""
    b.set_grid(None)
    b.set_stars(stars0)
    b.run_adabelief(ITERS_PER_STEP, **ab)


def effective_cpus():
    """Hardware threads this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a
    job a share of its host: os.cpu_count() says 256 there, the quota 16 - 256 OpenMP threads on 16 cores is what a barrier
    or a dynamic schedule pays for dearly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path, parse in (('/sys/fs/cgroup/cpu.max', lambda t: t.split()),):
        try:
            quota, period = parse(open(path).read())
            if quota != 'max':
                n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
        except Exception:
            pass
    try:   # cgroup v1
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            n = min(n, max(1, int(q / p + 0.5)))
    except Exception:
        pass
    return max(n, 1)


def cpu_baseline(ds, weight, b, stars0, ss, seconds_target=12.0):
    """oracle/psf_cpu.c (fp32 C + OpenMP over (frame, star) work units, the same algorithm as the HIP path; kind 'port') timed on the
    host cores on a bounded sample of the same C2 workload: all hardware threads, then one thread."""
    from oracle import model as om, psf_cpu
    try:  # compile for the CPU this runs on; the portable build shipped with the repo is the fall-back
        path = psf_cpu.build(native=True, out=os.path.join(tempfile.mkdtemp(prefix='lcmi_cpu_', dir='/tmp'), 'libpsfcpu.so'))
        handle = psf_cpu.lib(path)
    except Exception:
        handle = psf_cpu.lib()
    F, S, n, _ = ds['data'].shape
    N = n * ss
    mof = b.get_moffat()
    Tm = np.stack([om.moffat(N, ss, *[om.T(float(v)) for v in mof[f]]).numpy() for f in range(F)])
    W = b.get_weights()
    threads = effective_cpus()
    out = {}
    for label, thr, frames, iters in (('all', threads, F, 60), ('one', 1, min(F, 8), 20)):
        st = psf_cpu.PsfCpuState(ds['data'][:frames], weight[:frames], ss, Tm[:frames], W[:frames],
                                 np.zeros((frames, N * N), np.float32), stars0[:frames], handle)
        t0 = time.perf_counter()
        st.run_adabelief(5, threads=thr)
        rate = frames * S * 5 / (time.perf_counter() - t0)
        budget = seconds_target * (0.75 if label == 'all' else 0.25)
        iters = int(max(iters, min(3000, budget * rate / (frames * S))))
        t0 = time.perf_counter()
        hist = st.run_adabelief(iters, threads=thr)
        dt = time.perf_counter() - t0
        out[label] = dict(rate=frames * S * iters / dt, frames=frames, iters=iters, seconds=dt,
                          finite=bool(np.all(np.isfinite(hist))))
    return dict(value=out['all']['rate'], unit='cutouts/sec', cores=threads, host_logical_cpus=os.cpu_count(), kind='port',
                sample=f"{out['all']['frames']} frames x {S} stamps x {out['all']['iters']} AdaBelief iterations of the same "
                       f"C2 data ({out['all']['seconds']:.1f} s), oracle/psf_cpu.c fp32 + OpenMP (whole frames per thread; (frame, star) units when there are more threads than frames)",
                value_one_thread=out['one']['rate'],
                sample_one_thread=f"{out['one']['frames']} frames x {S} stamps x {out['one']['iters']} iterations "
                                  f"({out['one']['seconds']:.1f} s)",
                loss_finite=out['all']['finite'] and out['one']['finite'])


def joint_workload(ctx, E, n, M, seed, iters, label):
    """Joint ROI forward-model iterations (all parameters free, the reference's ROI regularisation strengths:
    roi_modelling.py:308-321) on this GPU; returns an entry with its own roofline block."""
    from lightcurver_amd.joint import JointFit
    from lightcurver_amd.synthetic import make_roi_dataset
    ss = 2
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=seed)
    j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, ctx)
    p = dict(ds['truth'])
    p['a'] = p['a'] * 0.9
    j.set_params(**p)
    t0 = time.perf_counter()
    W = j.propagate_noise()
    ctx.synchronize()
    setup_s = time.perf_counter() - t0
    j.set_loss(W=W, lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
    j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
    ab = dict(init_learning_rate=1e-4, schedule_learning_rate=False)   # roi_modelling.py:329
    j.run_adabelief(10, **ab)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.timer_start()
    j.run_adabelief(iters, **ab)
    ms = ctx.timer_stop()
    wall = time.perf_counter() - t0
    hist = j.loss_history()
    j.close()
    bytes_per = joint_bytes_per_cutout_iteration(n, ss, E)
    flops_per = joint_flops_per_cutout_iteration(n, ss)
    it_s = ms * 1e-3 / iters
    return {'workload': f'{label}: {E} epochs x {n}x{n} ROI, {M} point sources + background, all parameters free, '
                        f'reference-default regularisation, {iters} AdaBelief iterations',
            'value': E * iters / (ms * 1e-3), 'unit': 'cutouts/sec', 'us_per_iteration': it_s * 1e6,
            'wall_us_per_iteration': wall * 1e6 / iters, 'setup_seconds_untimed': setup_s,
            'roofline': hbm_roofline(E * bytes_per, it_s, 'one joint iteration: joint_epoch_kernel + reduction + update',
                                     {'algorithmic_bytes_per_cutout_iteration': bytes_per,
                                      'fp32_valu_tflops': E * flops_per / it_s / 1e12,
                                      'fp32_valu_frac_of_157': E * flops_per / it_s / 1e12 / FP32_PEAK_TFLOPS,
                                      # the executed-flop figure beside SURVEY's algorithmic one (the kernel transforms at
                                      # L = 1.5 N with binned rows: joint_flops_executed_per_cutout_iteration)
                                      'fp32_valu_tflops_executed': E * joint_flops_executed_per_cutout_iteration(n, ss) / it_s / 1e12,
                                      'fp32_valu_frac_of_157_executed': E * joint_flops_executed_per_cutout_iteration(n, ss) / it_s / 1e12 / FP32_PEAK_TFLOPS}),
            'loss_finite': bool(np.all(np.isfinite(hist))), 'loss_first_last': [float(hist[0]), float(hist[-1])]}


def joint_cpu_port(n, M, seed, epochs=32, seconds_target=8.0):
    """CPU figure beside a joint-fit entry (north_star: "next to the STARRED-CPU path timed on the same box's host cores"):
    oracle/joint_cpu.c - the joint fit WITH the background as plain C + OpenMP over the epochs (radix-2 FFT convolution, hand-derived
    adjoints, the same loss terms and AdaBelief as the HIP path; fp32 build; its fp64 build is pinned to oracle/model.py at 1e-9
    in tests/test_joint_cpu_port_cpu.py) - on `epochs` epochs of the same synthetic workload, full AdaBelief iterations, all the
    cores the job may use, then one.  The loop it stands for: lightcurver/processes/roi_modelling.py:308-334."""
    import tempfile as _tf
    from oracle import joint_cpu
    from lightcurver_amd.synthetic import make_roi_dataset
    try:  # compiled for the CPU this runs on; the portable build shipped with the repo is the fall-back
        path = joint_cpu.build(native=True, out=os.path.join(_tf.mkdtemp(prefix='lcmi_cpuj_', dir='/tmp'), 'libjointcpu.so'))
    except Exception:
        path = None
    ss = 2
    cores = effective_cpus()
    epochs = max(epochs, cores)
    ds = make_roi_dataset(E=epochs, M=M, n=n, ss=ss, seed=seed)
    out = {}
    for label, thr, budget in (('all', cores, 0.7 * seconds_target), ('one', 1, 0.3 * seconds_target)):
        c = joint_cpu.JointCpu(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, double=False, threads=thr,
                               lib_path=path)
        p = dict(ds['truth'])
        p['a'] = np.asarray(p['a']) * 0.9
        c.set_params(**p)
        c.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
        c.run(1, threads=thr)                      # thread pool, first touch of the work space
        k, t0 = 0, time.perf_counter()
        while True:
            hist = c.run(2, threads=thr)
            k += 3                                 # (a run of 2 iterations evaluates the model three times)
            dt = time.perf_counter() - t0
            if dt > budget or k >= 600:
                break
        c.close()
        out[label] = dict(rate=epochs * k / dt, k=k, dt=dt, finite=bool(np.all(np.isfinite(hist))))
    return dict(value=out['all']['rate'], unit='cutouts/sec', cores=cores, host_logical_cpus=os.cpu_count(), kind='port',
                sample=f"{out['all']['k']} forward + backward evaluations with AdaBelief updates on {epochs} epochs x {n}x{n} of the same "
                       f"synthetic workload ({out['all']['dt']:.1f} s), oracle/joint_cpu.c fp32 + OpenMP over the epochs",
                value_one_thread=out['one']['rate'],
                sample_one_thread=f"{out['one']['k']} evaluations on the same epochs ({out['one']['dt']:.1f} s)",
                loss_finite=out['all']['finite'] and out['one']['finite'])


def joint_cpu_oracle(n, M, seed, epochs=4, seconds_target=6.0):
    """CPU figure beside a joint-fit entry: loss + full gradient of the same model by the float64 torch oracle (oracle/model.py,
    autograd - the checker of the parity tests, kind 'oracle': not a tuned port) on `epochs` epochs of the same synthetic
    workload, all host threads torch uses.  One evaluation = one cutout-iteration per epoch without the optimiser update."""
    import torch
    from oracle import model as om, optim as oo
    from lightcurver_amd.synthetic import make_roi_dataset
    ss = 2
    ds = make_roi_dataset(E=epochs, M=M, n=n, ss=ss, seed=seed)
    p = {k: om.T(v) for k, v in ds['truth'].items()}
    p['a'] = p['a'] * 0.9
    data, sig2, psf = om.T(ds['data']), om.T(ds['noisemap']) ** 2, om.T(ds['psf'])
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h']
    fn = lambda q: om.deconv_loss(q, data, sig2, psf, ss, lam_scales=1.0, lam_hf=1.0, lam_pos=100.0, lam_pts=0.01, lam_fu=10.0)
    oo.value_and_grad(fn, p, free)
    t0 = time.perf_counter()
    k = 0
    while True:
        oo.value_and_grad(fn, p, free)
        k += 1
        dt = time.perf_counter() - t0
        if dt > seconds_target or k >= 200:
            break
    return dict(value=epochs * k / dt, unit='cutouts/sec', cores=min(torch.get_num_threads(), effective_cpus()), kind='oracle',
                sample=f'{k} evaluations of loss + full gradient on {epochs} epochs x {n}x{n} of the same synthetic workload '
                       f'({dt:.1f} s), oracle/model.py in torch float64 with autograd (the parity checker, not a tuned port)')


def distortion_workload(ctx, iters=300):
    """Pixel-grid stage of build_psf(field_distortion=True) (the mode the reference's integration test runs,
    tests/test_entire_pipeline/test_run_pipeline_example_config.py:113-128) at C2's size: per iteration the pixel grid is
    resampled for every star, the stars step, the adjoint resampling sums their gradients and the grid steps - four
    launches per iteration, the loop in the library (lc_psf_distortion_run)."""
    from lightcurver_amd.psf_batch import PsfBatch
    from lightcurver_amd.starred.procedures.psf_routines import quadratic_forms
    from lightcurver_amd.synthetic import make_psf_dataset
    F, S, n, ss = 100, 8, 32, 2
    ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=102)
    rng = np.random.default_rng(7)
    weight = (ds['masks'] / ds['noisemap'].astype(np.float64) ** 2).astype(np.float32)
    g = ds['fwhm_guess']
    f0 = np.sqrt(np.maximum(g * g - (2.0 / ss) ** 2, 1.0))
    theta = np.zeros((F, 13))
    theta[:, 0], theta[:, 1], theta[:, 3] = f0, f0, 2.5
    theta[:, 4:13] = rng.uniform(-0.02, 0.02, (F, 9))         # a field distortion of a few per cent
    xy = rng.uniform(-0.5, 0.5, (F, S, 2))
    base = PsfBatch(ds['data'], weight, ss, ctx)
    stars = np.zeros((F, S, 4), np.float32)
    stars[..., 0] = (ds['data'] * ds['masks']).sum(axis=(-1, -2))
    base.set_moffat(theta[:, 0:4])
    base.set_stars(stars)
    base.set_grid(None)
    base.propagate_noise()
    W = base.get_weights()
    base.close()
    star_b = PsfBatch(ds['data'].reshape(F * S, 1, n, n), weight.reshape(F * S, 1, n, n), ss, ctx)
    frame_b = PsfBatch(np.zeros((F, 1, n, n), np.float32), np.zeros((F, 1, n, n), np.float32), ss, ctx)
    star_b.set_grid(None)
    star_b.set_moffat_q(quadratic_forms(theta, xy, ss).reshape(F * S, 4))
    star_b.set_stars(stars.reshape(F * S, 1, 4))
    star_b.set_regularization(None, 0.0, 0.0)
    frame_b.set_moffat(theta[:, 0:4])
    frame_b.set_grid(None)
    frame_b.set_regularization(W, 1.0, 1.0)
    frame_b.set_distortion(S, theta[:, 4:13], xy)
    ab = dict(init_learning_rate=1e-4, schedule_learning_rate=True)
    frame_b.distortion_run(star_b, 10, **ab)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.timer_start()
    frame_b.distortion_run(star_b, iters, **ab)
    ms = ctx.timer_stop()
    wall = time.perf_counter() - t0
    hist = frame_b.loss_history() + star_b.loss_history().reshape(F, S, -1).sum(axis=1)
    star_b.close()
    frame_b.close()
    return {'workload': f'C2 with field_distortion=True: {F} frames x {S} stars, {n}x{n} stamps, pixel-grid stage, '
                        f'{iters} AdaBelief iterations, four launches per iteration driven from C++ (lc_psf_distortion_run)',
            'value': F * S * iters / (ms * 1e-3), 'unit': 'cutouts/sec', 'us_per_iteration': ms * 1e3 / iters,
            'wall_us_per_iteration': wall * 1e6 / iters, 'loss_finite': bool(np.all(np.isfinite(hist))),
            'loss_first_last': [float(hist[:, 0].sum()), float(hist[:, -1].sum())]}


def c3_shard_workload(ctx, iters=200):
    """One GPU's share of BASELINE.json configs[2] (C3: 500 frames x 8 stars x 64x64 over 8 GPUs = 63 frames)."""
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
    b.close()
    bytes_per = psf_bytes_per_cutout_iteration(n, ss, S)
    N = n * ss
    return {'workload': f'C3 shard: {F} frames x {S} stars, {n}x{n} stamps (1/8 of C3), {iters} AdaBelief iterations, one launch',
            'value': F * S * iters / (ms * 1e-3), 'unit': 'cutouts/sec', 'us_per_iteration': ms * 1e3 / iters,
            'roofline': hbm_roofline(F * S * iters * bytes_per, ms * 1e-3, 'psf_fit_kernel (N = 128)',
                                     {'algorithmic_bytes_per_cutout_iteration': bytes_per, 'iterations_per_figure': iters,
                                      'fp32_valu_frac_of_157': F * S * iters * 58.5 * N * N / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}),
            'loss_finite': bool(np.all(np.isfinite(hist)))}


def c3_sharded_fit(ctx, rank, world, dist, iters=200):
    """BASELINE.json configs[2] as north_star splits it: C3's 500 frames x 8 stars x 64 x 64 sharded over the ranks,
    ceil(500 / N) frames each (the last rank takes what is left), no data-path collective - the frames are independent fits
    (reference: the serial loop over frames at lightcurver/processes/psf_modelling.py:92).  Strong scaling: the 500 frames
    are fixed; the time is the slowest rank's."""
    import torch
    from lightcurver_amd.psf_batch import PsfBatch
    from lightcurver_amd.synthetic import make_psf_dataset
    Ftot, S, n, ss = 500, 8, 64, 2
    per = (Ftot + world - 1) // world
    lo, hi = min(rank * per, Ftot), min((rank + 1) * per, Ftot)
    F = hi - lo
    ms = 0.0
    finite = True
    if F > 0:
        ds = make_psf_dataset(F=F, S=S, n=n, ss=ss, seed=103 + 1000 * rank)
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
    dist.barrier()
    t0 = time.perf_counter()
    if F > 0:
        b.run_adabelief(iters, **ab)
        ctx.synchronize()
    dt = time.perf_counter() - t0
    if F > 0:
        finite = bool(np.all(np.isfinite(b.loss_history())))
        b.close()
    t = torch.tensor([dt, 0.0 if finite else 1.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0])
    bytes_per = psf_bytes_per_cutout_iteration(n, ss, S)
    return {'workload': f'C3 sharded: {Ftot} frames x {S} stars, {n}x{n} stamps over {world} ranks ({per} frames per rank, no '
                        f'collective), {iters} AdaBelief iterations, one launch per rank',
            'value': Ftot * S * iters / dt, 'unit': 'cutouts/sec', 'us_per_iteration': dt * 1e6 / iters, 'scaling': 'strong',
            'frames_per_rank': per, 'ranks': world,
            'roofline_frac_per_gpu': Ftot * S * iters * bytes_per / dt / 1e9 / HBM_PEAK_GBS / world,
            'loss_finite': float(t[1]) == 0.0}


def star_photometry_workload(ctx, iters=2000, with_cpu=True):
    """The reference's star photometry (star_photometry.py:257-326: 30 reference stars, each a 2000-iteration joint fit of
    one point source over its epochs, background fixed at zero) as one batched device fit: 30 stars x 100 epochs x 32x32,
    fluxes, star position and per-epoch shifts free.  CPU beside it: oracle/joint_ps_cpu.c (fp32 C + OpenMP over epochs, the
    same separable algorithm) on one star's 100 epochs."""
    from lightcurver_amd.joint import StarPhotometryBatch
    from lightcurver_amd.synthetic import make_roi_dataset
    G, E, n, ss = 30, 100, 32, 2
    base = make_roi_dataset(E=E, M=1, n=n, ss=ss, seed=106, with_background=False)
    sig2 = base['noisemap'].astype(np.float64) ** 2
    b = StarPhotometryBatch([(base['data'], sig2, base['psf'])] * G, ss, 1, ctx)
    a0 = np.asarray(base['truth']['a'], np.float64) * 0.9
    b.set_params(a=np.tile(a0, G), c_x=np.zeros(G), c_y=np.zeros(G), dx=np.zeros(G * E), dy=np.zeros(G * E),
                 alpha=np.zeros(G * E), mean=np.zeros(G * E))
    b.set_loss()
    b.set_free(['a', 'c_x', 'c_y', 'dx', 'dy'])
    ab = dict(init_learning_rate=1e-3, schedule_learning_rate=True)
    b.run_adabelief(5, **ab)
    ctx.synchronize()
    ctx.timer_start()
    b.run_adabelief(iters, **ab)
    ms = ctx.timer_stop()
    hist = b.loss_history()
    b.close()
    N = n * ss
    bytes_per = 8 * n * n + 4 * N * N     # data + 1/sigma^2 + the epoch's narrow PSF (no background term)
    out = {'workload': f'star photometry: {G} stars x {E} epochs x {n}x{n} in one batched fit (lc_joint_create_groups), '
                       f'{iters} AdaBelief iterations',
           'value': G * E * iters / (ms * 1e-3), 'unit': 'cutouts/sec', 'us_per_iteration': ms * 1e3 / iters,
           'roofline': hbm_roofline(G * E * iters * bytes_per, ms * 1e-3, 'joint_ps_kernel + joint_update_groups_kernel',
                                    {'algorithmic_bytes_per_cutout_iteration': bytes_per, 'iterations_per_figure': iters}),
           'loss_finite': bool(np.all(np.isfinite(hist)))}
    if with_cpu:
        try:
            from oracle.joint_ps_cpu import JointPsCpu
            cores = effective_cpus()
            c = JointPsCpu(base['data'], sig2, base['psf'], ss, 1, double=False)
            c.set_params(a=a0)
            c.run(5, threads=cores)
            k, t0 = 0, time.perf_counter()
            while True:
                c.run(50, threads=cores)
                k += 50
                dt = time.perf_counter() - t0
                if dt > 5.0 or k >= 4000:
                    break
            out['cpu_baseline'] = dict(value=E * k / dt, unit='cutouts/sec', cores=cores, kind='port',
                                       sample=f'{k} AdaBelief iterations of one star ({E} epochs x {n}x{n}, the same synthetic '
                                              f'stack) in {dt:.1f} s: oracle/joint_ps_cpu.c, fp32 C + OpenMP over the epochs')
        except Exception as e:
            out['cpu_baseline'] = {'value': None, 'error': repr(e)}
    return out


def all_ranks_agree(step, fn, world):
    """Runs fn on this rank, then lets the ranks agree (host-side default group): a failure anywhere raises everywhere, so
    that no rank walks into a collective or a barrier the failed one never reaches (tests/test_bench_supervisor_cpu.py
    rehearses it on eight gloo ranks)."""
    import torch.distributed as dist
    err, out = None, None
    try:
        out = fn()
    except Exception as e:
        err = repr(e)
    errs = [None] * world
    dist.all_gather_object(errs, err)
    bad = [f'rank {r}: {e}' for r, e in enumerate(errs) if e]
    if bad:
        raise RuntimeError(f'{step}: ' + '; '.join(bad))
    return out


def sharded_joint_fit(ctx, rank, world, iters=500, transport='rccl', config='C4'):
    """C4's 200 epochs (or, config='C5', BASELINE.json configs[4]: 1000 epochs of 128 x 128 with 4 sources, the joint fit
    that needs the eight GPUs) sharded over the ranks; the shared block is all-reduced every iteration - in place by RCCL
    through the library's own communicator (transport 'rccl': lc_rccl_allreduce is the callback of the C++ loop, no Python
    per iteration; in the one-GPU rehearsal, where RCCL refuses two ranks on one device, gloo staged through the host), by
    torch.distributed's nccl group called back from the loop ('torch-nccl', the cross-check) or by the library's one-shot
    peer-memory kernel over HIP IPC ('peer', csrc/peer.hip); the loop runs in C++ (lc_joint_run_sharded).
    Strong scaling: the total work is fixed."""
    import datetime
    import torch
    import torch.distributed as dist
    from lightcurver_amd.distributed import PeerGroup, RcclGroup, ShardedJointOptimizer, shard_epochs
    from lightcurver_amd.joint import JointFit
    from lightcurver_amd.synthetic import make_roi_dataset
    E, n, M, ss, seed = (1000, 128, 4, 2, 105) if config == 'C5' else (200, 64, 2, 2, 104)
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=seed)  # same seed on every rank: identical full problem
    lo, hi = shard_epochs(E, world, rank)

    def all_ranks(step, fn):
        return all_ranks_agree(step, fn, world)

    def make_fit():
        j = JointFit(ds['data'][lo:hi], ds['noisemap'][lo:hi].astype(np.float64) ** 2, ds['psf'][lo:hi], ss, M, ctx)
        p = dict(ds['truth'])
        p['a'] = (np.asarray(p['a']).reshape(E, M) * 0.9)[lo:hi].reshape(-1)
        for k in ('dx', 'dy', 'alpha', 'mean'):
            p[k] = np.asarray(p[k])[lo:hi]
        j.set_params(**p)
        j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=100.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
        j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
        return j

    j = all_ranks('set-up of the local fit', make_fit)
    peer = rccl = None
    group = None
    if transport == 'peer':
        peer = PeerGroup(j)                 # IPC handles travel over the default (gloo) group; fails on every rank or on none
    elif os.environ.get('LCMI_BENCH_DEVICE') is not None:
        # one-GPU rehearsal (every rank on the same device): RCCL refuses two ranks on one GPU, so the shared block
        # is staged through the host over the gloo group
        pass
    elif transport == 'torch-nccl':
        torch.cuda.set_device(ctx.stream()[1])
        group = dist.new_group(backend='nccl', timeout=datetime.timedelta(seconds=180))
    else:
        rccl = RcclGroup(ctx)               # the unique id travels over the default (gloo) group; raises on every rank or on none
    opt = ShardedJointOptimizer(j, group, peer=peer, rccl=rccl)
    ab = dict(init_learning_rate=1e-4, schedule_learning_rate=False)

    def run_synced(n):
        opt.run(n, **ab)
        ctx.synchronize()

    all_ranks('first iterations', lambda: run_synced(10))
    t0 = time.perf_counter()
    all_ranks('timed iterations', lambda: run_synced(iters))   # the agreement doubles as the closing barrier
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0])
    hist = j.loss_history()
    kind = opt.transport
    rccl_calls = rccl.calls if rccl is not None else 0
    if peer is not None:
        peer.close()
    if rccl is not None:
        rccl.close()
    j.close()
    transport = {'peer': 'one-shot peer-memory all-reduce (HIP IPC, every rank reads the others directly)',
                 'rccl-native': "RCCL all-reduce in place in device memory by the library's own communicator (no Python in the loop)",
                 'rccl': 'RCCL all-reduce in place in device memory (torch.distributed nccl group called back from the loop)',
                 'gloo': 'gloo all-reduce staged through the host (one-GPU rehearsal)'}[kind]
    return {'workload': f'{config} sharded: {E} epochs x {n}x{n} ROI over {world} ranks, {transport} of the shared block '
                        f'({n * ss * n * ss + 4 * M + 2} floats) once per iteration, {iters} iterations',
            'value': E * iters / dt, 'unit': 'cutouts/sec', 'us_per_iteration': dt * 1e6 / iters, 'scaling': 'strong',
            # what RCCL saw: the size of the nccl group the block was reduced over, 0 when the collective was gloo's
            'rccl_ranks': world if kind == 'rccl-native' else (dist.get_world_size(group) if kind == 'rccl' else 0),
            'rccl_all_reduces': rccl_calls, 'collective': kind,
            'device_collective': kind in ('rccl', 'rccl-native', 'peer'), 'ranks': world, 'loop': 'lc_joint_run_sharded (C++)',
            'loss_finite': bool(np.all(np.isfinite(hist)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)   # 50 steps of ~40 ms: a timed region of about two seconds
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='C2', choices=['C1', 'C2', 'C3'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', '--no-joint', dest='no_extra', action='store_true',
                    help='skip the C4 / C5-shard / C3-shard entries')
    ap.add_argument('--no-traffic', action='store_true', help='skip the rocprofv3 --pmc passes (roofline.traffic = null)')
    ap.add_argument('--no-sharded-joint', action='store_true', help='N > 1: skip the epoch-sharded C4 fit with RCCL')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--pmc-child-sections', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child_sections:
        pmc_sections_child()
        return

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.gpus > 1 and os.environ.get('LCMI_BENCH_WORKER') != '1' and not args.no_sharded_joint:
        sys.exit(supervise_rank(sys.argv[1:]))

    # stdout carries ONE line, the JSON of rank 0: libraries that chat on file descriptor 1 (gloo's "[Gloo] Rank 0 is connected
    # to ..." from C++) are sent to stderr for the rest of the run; the line itself is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != max(args.gpus, 1):
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)
        sys.exit(2)

    traffic, traffic_source = None, 'not measured (--no-traffic, N > 1 or a profiler child)'
    joint_traffic = {}
    if world == 1 and not args.no_traffic and not args.pmc_child and args.config == 'C2':
        traffic, traffic_source = measure_traffic()   # before this process touches the GPU
        if not args.no_extra:
            joint_traffic = measure_section_traffic()

    dist = None
    if world > 1:
        import torch.distributed as dist
        # frames shard with no data-path collective: this host-side (gloo) group only carries the barriers and the
        # max-over-ranks of the timing; the sharded joint fit makes its own nccl (= RCCL) group
        import datetime
        # bounded: a rank that died makes the others raise after five minutes instead of waiting for half an hour
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))

    from lightcurver_amd import _lib
    # rehearsal of the multi-rank path on a one-GPU box: LCMI_BENCH_DEVICE=0 puts every rank on that device
    ctx = _lib.Context(int(os.environ.get('LCMI_BENCH_DEVICE', local_rank)))
    ds, weight, b, stars0, setup_s, (F, S, n, ss) = setup_psf_batch(ctx, args.config, rank)
    ab = dict(init_learning_rate=1e-4, schedule_learning_rate=True)

    if args.pmc_child:   # under rocprofv3 --pmc: two launches of the timed kind, nothing else
        for _ in range(2):
            psf_step(b, stars0, ab)
        ctx.synchronize()
        return

    for _ in range(args.warmup):
        psf_step(b, stars0, ab)
    ctx.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(args.steps):
        b.set_grid(None)
        b.set_stars(stars0)
        ctx.timer_start()                       # HIP events on the stream the kernel runs on
        b.run_adabelief(ITERS_PER_STEP, **ab)
        kernel_ms += ctx.timer_stop()           # also synchronises
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

    if rank == 0:
        value = F * S * world * ITERS_PER_STEP * args.steps / elapsed
        bytes_per = psf_bytes_per_cutout_iteration(n, ss, S)
        launch_s = kernel_ms * 1e-3 / args.steps
        N = n * ss
        flops_per = 58.5 * N * N  # separable 13-tap passes: 29.25 N^2 multiply-adds per stamp-iteration
        roof = hbm_roofline(F * S * ITERS_PER_STEP * bytes_per, launch_s, 'psf_fit_kernel',
                            {'algorithmic_bytes_per_cutout_iteration': bytes_per,
                             'fp32_valu_tflops': F * S * ITERS_PER_STEP * flops_per / launch_s / 1e12,
                             'fp32_valu_frac_of_157': F * S * ITERS_PER_STEP * flops_per / launch_s / 1e12 / FP32_PEAK_TFLOPS,
                             'note': 'the pixel state stays on chip across the iterations of a launch, so HBM moves far '
                                     'fewer bytes than the algorithmic figure; at 2 waves / SIMD the kernel is bound by '
                                     'instruction issue and the latencies it cannot hide (SQ counters, profiles/: waves '
                                     'parked 40 %, issuing 42 %, of which VALU 31 %)'})
        roof['traffic'] = traffic
        roof['traffic_source'] = traffic_source
        out = {
            'metric': 'cutouts/sec (PSF-fit + joint forward-model iter), 32x32 & 64x64 stamps',
            'value': value, 'unit': 'cutouts/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.config}: {F} frames x {S} stars, {n}x{n} stamps, subsampling {ss}, '
                                   f'starlet-regularised PSF pixel-grid fit; one step = the whole stage of '
                                   f'{ITERS_PER_STEP} AdaBelief iterations from B = 0 (one persistent launch)',
                       'frames_per_gpu': F, 'stars': S, 'stamp': n, 'subsampling': ss,
                       'iters_per_step': ITERS_PER_STEP, 'us_per_iteration': launch_s * 1e6 / ITERS_PER_STEP,
                       'sharding': f'frames x{world} (no collective)',
                       'setup_seconds_untimed': setup_s, 'loss_finite': finite,
                       'median_reduced_chi2': float(np.median(res['chi2']))},
            'roofline': roof,
        }
        try:  # the box's own copy bandwidth beside the 8 TB/s specification
            import ctypes
            g = ctypes.c_float()
            ctx.check(_lib.lib().lc_copy_bandwidth(ctx.h, 1 << 30, 10, ctypes.byref(g)), 'lc_copy_bandwidth')
            out['roofline']['measured_copy_GBps'] = g.value
            out['roofline']['frac_of_measured_copy'] = roof['achieved'] / g.value
        except Exception:
            out['roofline']['measured_copy_GBps'] = None
        if world == 1 and not args.no_extra:
            extra = []
            for fn, kw in ((joint_workload, dict(E=200, n=64, M=2, seed=104, iters=2000, label='C4')),
                           (joint_workload, dict(E=25, n=64, M=2, seed=104, iters=2000,
                                                 label="C4 shard (one GPU's eighth of C4's 200 epochs)")),
                           (joint_workload, dict(E=125, n=128, M=4, seed=105, iters=200,
                                                 label="C5 shard (one GPU's eighth of C5's 1000 epochs)")),
                           (c3_shard_workload, {}),
                           (distortion_workload, {}),
                           (star_photometry_workload, dict(with_cpu=not args.no_cpu_baseline))):
                try:
                    extra.append(fn(ctx, **kw))
                except Exception as e:
                    extra.append({'workload': kw.get('label', fn.__name__), 'error': repr(e)})
            # counter traffic of one iteration (one launch-iteration of the C3 shard), measured before this process touched the GPU;
            # per launch like `achieved`: the C3-shard and star-photometry entries time `iters` iterations in their figure
            for w in extra:
                label = w.get('workload', '')
                key = next((k for k in sorted(joint_traffic, key=len, reverse=True) if label.startswith(k + ':') or label.startswith(k + ' (')), None)
                if key is not None and 'roofline' in w:
                    tr, src = joint_traffic[key]
                    per_launch = w['roofline'].pop('iterations_per_figure', 1)
                    w['roofline']['traffic'] = None if tr is None else tr * per_launch
                    w['roofline']['traffic_source'] = src + (f' (x {per_launch} iterations of the timed figure)' if per_launch != 1 else '')
            if not args.no_cpu_baseline:   # the CPU path timed beside the joint fits too (north_star): C4 and the C5 shard
                for idx, kw in ((0, dict(n=64, M=2, seed=104, seconds_target=8.0)), (2, dict(n=128, M=4, seed=105, seconds_target=6.0))):
                    try:
                        extra[idx]['cpu_baseline'] = joint_cpu_port(**kw)
                    except Exception as e:
                        extra[idx]['cpu_baseline'] = {'value': None, 'error': repr(e)}
            out['config']['other_workloads'] = extra
        if not args.no_cpu_baseline and world == 1:
            try:
                out['cpu_baseline'] = cpu_baseline(ds, weight, b, stars0, ss)
            except Exception as e:  # the bench line must still be printed
                out['cpu_baseline'] = {'value': None, 'error': repr(e)}

    # N > 1: the headline is complete.  Under the supervisor (supervise_rank) rank 0 hands over a provisional line and every
    # rank the marker, so that nothing in the sharded joint fits below - the part only a multi-GPU node can run - can
    # take the headline with it.
    worker = os.environ.get('LCMI_BENCH_WORKER') == '1'
    if world > 1 and worker:
        if rank == 0:
            os.write(real_stdout, (json.dumps(out) + '\n').encode())
        os.write(real_stdout, b'HEADLINE_DONE\n')

    sharded = sharded_peer = sharded_c5 = sharded_c5_rccl = sharded_c3 = None
    if world > 1 and not args.no_sharded_joint:
        try:   # C3 as north_star splits it: the 500 frames over the ranks, no collective
            sharded_c3 = c3_sharded_fit(ctx, rank, world, dist)
        except Exception as e:
            sharded_c3 = {'error': repr(e)}
        try:
            sharded = sharded_joint_fit(ctx, rank, world)
        except Exception as e:
            sharded = {'error': repr(e)}
        try:   # the same fit with the one-shot peer-memory all-reduce instead of the collective
            sharded_peer = sharded_joint_fit(ctx, rank, world, transport='peer')
        except Exception as e:
            sharded_peer = {'error': repr(e)}
        try:   # C5, the configuration sharding is for: 1000 epochs of 128 x 128 (one GPU alone: ~1.54 ms per iteration)
            sharded_c5_rccl = sharded_joint_fit(ctx, rank, world, iters=100, transport='rccl', config='C5')
        except Exception as e:
            sharded_c5_rccl = {'error': repr(e)}
        try:
            sharded_c5 = sharded_joint_fit(ctx, rank, world, iters=100, transport='peer', config='C5')
        except Exception as e:
            sharded_c5 = {'error': repr(e)}

    if rank == 0:
        if sharded is not None:
            out['config']['sharded_joint_fit'] = sharded
            out['config']['rccl_ranks'] = sharded.get('rccl_ranks')
        if sharded_peer is not None:
            out['config']['sharded_joint_fit_peer'] = sharded_peer
        if sharded_c5 is not None:
            out['config']['sharded_joint_fit_c5_peer'] = sharded_c5
        if sharded_c5_rccl is not None:
            out['config']['sharded_joint_fit_c5'] = sharded_c5_rccl
        if sharded_c3 is not None:
            out['config']['sharded_psf_fit_c3'] = sharded_c3
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
