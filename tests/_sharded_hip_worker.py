"""One rank of the two-process HIP test (tests/test_distributed_gpu.py): fits its contiguous block of the epochs on GPU 0
through lightcurver_amd.distributed.ShardedJointOptimizer with a real JointFit object, the shared block summed over the
ranks by a gloo all-reduce (both ranks share the one GPU of the test box, where RCCL refuses to run two ranks) or by the
library's one-shot peer-memory all-reduce over HIP IPC (transport 'peer'); the loop itself runs in C++ (lc_joint_run_sharded).
usage: RANK=r WORLD_SIZE=w MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/_sharded_hip_worker.py out.npz E M n T [gloo|peer]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, E, M, n, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    transport = sys.argv[6] if len(sys.argv) > 6 else 'gloo'   # 'peer': the one-shot peer-memory all-reduce (HIP IPC)
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lightcurver_amd import _lib
    from lightcurver_amd.distributed import PeerGroup, ShardedJointOptimizer, gather_epoch_blocks, shard_epochs, shard_kwargs
    from lightcurver_amd.joint import JointFit
    from lightcurver_amd.synthetic import make_roi_dataset
    ss = 2
    ds = make_roi_dataset(E=E, M=M, n=n, ss=ss, seed=4242)
    p = {k: np.asarray(v, dtype=np.float64) for k, v in ds['truth'].items()}
    p['a'] = p['a'] * 0.9
    lo, hi = shard_epochs(E, world, rank)
    ctx = _lib.Context(0)
    if transport.startswith('roi'):   # 'roi' / 'roi-peer': the whole two-stage fit, sharded (processes/roi_modelling.py)
        from lightcurver_amd.processes.roi_modelling import global_scale, initial_point_source_fluxes, model_roi_cutouts_sharded
        off = (n - 1) / 2.0
        xs, ys = np.asarray(ds['truth']['c_x']) + off, np.asarray(ds['truth']['c_y']) + off
        scale = global_scale(ds['data'][lo:hi])
        a0 = initial_point_source_fluxes(ds['data'] / scale, xs, ys, 3.0)     # (every rank from the full stack: same numbers)
        res = model_roi_cutouts_sharded(ds['data'][lo:hi], ds['noisemap'][lo:hi], ds['psf'][lo:hi], ss, xs, ys,
                                        np.asarray(a0) * scale, scale, use_peer=(transport == 'roi-peer'),
                                        roi_deconv_translations_iters=T, roi_deconv_all_iters=300, ctx=ctx)
        if rank == 0:
            np.savez(out, hist=res['loss_history'], hist1=res['loss_history_stage1'], sigma=res['fluxes_sigma'], scale=res['scale'],
                     **{'p_' + k: v for k, v in res['flat_final'].items()})
        dist.barrier()
        dist.destroy_process_group()
        return
    j = JointFit(ds['data'][lo:hi], ds['noisemap'][lo:hi].astype(np.float64) ** 2, ds['psf'][lo:hi], ss, M, ctx)
    j.set_params(**shard_kwargs(p, E, M, world, rank))
    j.set_loss(lam_scales=1.0, lam_hf=1.0, lam_positivity=10.0, lam_pts_source=0.01, lam_flux_uniformity=10.0)
    j.set_free(['a', 'c_x', 'c_y', 'dx', 'dy', 'mean', 'h'])
    lbfgs = transport.startswith('lbfgs')     # 'lbfgs' / 'lbfgs-peer': the sharded L-BFGS-B stage instead of the AdaBelief loop
    peer = PeerGroup(j) if transport in ('peer', 'lbfgs-peer') else None
    opt = ShardedJointOptimizer(j, peer=peer)
    if lbfgs:
        from lightcurver_amd.distributed import sharded_lbfgs
        hist_l, res = sharded_lbfgs(opt, ['a', 'c_x', 'c_y', 'dx', 'dy'], T, lower={'a': 0.0})
        ctx.synchronize()
        full = gather_epoch_blocks(j.get_params(), M)
        if rank == 0:
            np.savez(out, hist=hist_l, fun=float(res.fun), nit=int(res.nit), **{'p_' + k: v for k, v in full.items()})
        dist.barrier()
        if peer is not None:
            peer.close()
        j.close()
        dist.destroy_process_group()
        return
    opt.run(T // 2, init_learning_rate=1e-3)
    opt.run(T - T // 2, init_learning_rate=1e-3)     # a second run continues the first (flux reference agreed again)
    ctx.synchronize()
    hist = j.loss_history()
    full = gather_epoch_blocks(j.get_params(), M)
    ref = j.get_flux_reference()
    if rank == 0:
        np.savez(out, hist=hist, flux_reference=ref, device_collective=bool(opt._dev), transport=opt.transport,
                 **{'p_' + k: v for k, v in full.items()})
    dist.barrier()
    if peer is not None:
        peer.close()
    j.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
