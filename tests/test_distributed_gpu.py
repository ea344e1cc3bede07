"""Device-resident collective of the sharded joint fit on the one GPU a test box has: a world-size-1 RCCL
group exercises the whole in-place path (raw device pointer viewed by torch, all-reduce enqueued on the
library's HIP stream, no host staging).  The N > 1 arithmetic is covered by the gloo test in
tests/test_host_logic_cpu.py; with one rank the all-reduce is the identity, so the sharded driver must
reproduce lc_joint_run_adabelief bit for bit."""
import socket

import numpy as np
import pytest

from lightcurver_amd.synthetic import make_roi_dataset

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_rccl_in_place_all_reduce_world_size_1(ctx):
    import torch
    import torch.distributed as dist
    from lightcurver_amd.distributed import ShardedJointOptimizer
    from lightcurver_amd.joint import JointFit
    E, M, n, ss, T = 6, 2, 32, 2, 12
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=4242)
    p = dict(ds['truth'])
    p['a'] = np.asarray(p['a']) * 0.9
    free = ['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h']
    fits = []
    for _ in range(2):
        j = JointFit(ds['data'], ds['noisemap'].astype(np.float64) ** 2, ds['psf'], ss, M, ctx)
        j.set_params(**p)
        j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0, lam_flux_uniformity=0.5)
        j.set_free(free)
        fits.append(j)
    fits[0].run_adabelief(T, init_learning_rate=1e-3)
    ref_hist = fits[0].loss_history()
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1)
    try:
        opt = ShardedJointOptimizer(fits[1])
        assert opt._device_collective(), 'the RCCL path must be taken with the nccl backend'
        opt.run(T, init_learning_rate=1e-3)
        ctx.synchronize()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    hist = fits[1].loss_history()
    np.testing.assert_array_equal(hist, ref_hist)
    a0, a1 = fits[0].get_params()['a'], fits[1].get_params()['a']
    np.testing.assert_array_equal(a0, a1)
    np.testing.assert_array_equal(fits[0].get_params()['h'], fits[1].get_params()['h'])
