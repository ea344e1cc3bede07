"""Epoch sharding of the joint fit over the GPUs of one node (one process per GPU).

Per-epoch parameters (a, dx, dy, mean) and their optimiser state live only on the rank that owns the
epoch; the shared parameters (h, c_x, c_y) are replicated.  Each iteration every rank runs the
forward/backward of its epochs, the shared block
    [ dL/dh (N^2) | dL/dc_x (M) | dL/dc_y (M) | sum_e (a - ref) (M) | sum_e (a - ref)^2 (M) | chi2 | n_epochs ]
is sum-all-reduced, and every rank applies the identical regularisation + AdaBelief update, so the
replicas stay in lock step.  The reference has no counterpart: it keeps all epochs in
one JAX array on one device (lightcurver/processes/roi_modelling.py:154-160,213).

With the nccl (= RCCL) backend the block is all-reduced in place in device memory, enqueued on the
library's own HIP stream (no host synchronisation inside the loop); with gloo (CPU tests) it is staged
through the host.  The PSF fit needs no collective at all (frames shard, see bench.py).
"""
import numpy as np


class _DeviceBlock:
    """Exposes a raw device pointer through __cuda_array_interface__ so that torch can view it."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = dict(shape=(int(count),), typestr='<f4', data=(int(ptr), False),
                                             version=2, strides=None)


def shard_epochs(n_epochs, world_size, rank):
    """Contiguous block of epochs owned by ``rank``: returns (start, stop)."""
    if not 0 <= rank < world_size:
        raise ValueError('rank out of range')
    base, rem = divmod(int(n_epochs), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_kwargs(flat, n_epochs, n_sources, world_size, rank):
    """Slice the per-epoch blocks of a flat parameter dict (a is epoch-major) for one rank."""
    lo, hi = shard_epochs(n_epochs, world_size, rank)
    out = dict(flat)
    out['a'] = np.asarray(flat['a']).reshape(n_epochs, n_sources)[lo:hi].reshape(-1)
    for k in ('dx', 'dy', 'alpha', 'mean'):
        out[k] = np.asarray(flat[k])[lo:hi]
    return out


class ShardedJointOptimizer:
    """Drives ``n_iter`` AdaBelief iterations of a sharded joint fit.

    ``local_fit`` is the rank's device object (``lightcurver_amd.joint.JointFit`` over the local epochs) or
    anything with the same four methods: step_local(), shared_get() -> float array, shared_set(array),
    step_update(**adabelief_cfg).
    """

    def __init__(self, local_fit, group=None):
        self.fit = local_fit
        self.group = group
        self._dev = None  # (tensor view of the shared block, torch ExternalStream of the library's stream)
        self._ref_agreed = False

    def _agree_flux_reference(self):
        """The flux moments of the shared block are centred on one reference flux per source (include/lcmi.h):
        every rank must use the same one, so the epoch-weighted mean of the local references is all-reduced (M + 1 numbers,
        at the start of every run)."""
        self._ref_agreed = True
        if not (hasattr(self.fit, 'get_flux_reference') and hasattr(self.fit, 'set_flux_reference')):
            return
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        E = float(self.fit.E)
        local = np.concatenate([np.asarray(self.fit.get_flux_reference(), np.float64) * E, [E]])
        if dist.get_backend(self.group) == 'nccl':
            _, device = self.fit.ctx.stream()
            t = torch.tensor(local, dtype=torch.float64, device=torch.device('cuda', device))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            tot = t.cpu().numpy()
        else:
            t = torch.from_numpy(local)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            tot = t.numpy()
        self.fit.set_flux_reference((tot[:-1] / tot[-1]).astype(np.float32))

    def _device_collective(self):
        """RCCL path: available when the process group is nccl and the fit is a device object."""
        import torch
        import torch.distributed as dist
        if self._dev is not None:
            return self._dev
        if not (dist.is_initialized() and dist.get_backend(self.group) == 'nccl'
                and hasattr(self.fit, 'shared_buffer') and hasattr(self.fit, 'ctx')):
            self._dev = False
            return False
        ptr, count = self.fit.shared_buffer()
        stream_ptr, device = self.fit.ctx.stream()
        view = torch.as_tensor(_DeviceBlock(ptr, count), device=torch.device('cuda', device))
        ext = torch.cuda.ExternalStream(stream_ptr, device=torch.device('cuda', device))
        self._dev = (view, ext)
        return self._dev

    def all_reduce_device(self):
        """Sum-all-reduce the shared block where it lives, ordered on the library's stream."""
        import torch
        import torch.distributed as dist
        view, ext = self._dev
        with torch.cuda.stream(ext):
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce(self, buf):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return buf
        t = torch.from_numpy(np.ascontiguousarray(buf))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.numpy()

    def run(self, n_iter, **adabelief_cfg):
        on_device = bool(self._device_collective())
        # agreed at the start of EVERY run: set_params(a=...) between two runs resets the local reference to the local
        # mean (lc_joint_set_param), and ranks centring their flux moments on different references corrupt the reduced sums
        self._agree_flux_reference()
        for _ in range(int(n_iter)):
            self.fit.step_local()
            if on_device:
                self.all_reduce_device()
            else:
                self.fit.shared_set(self.all_reduce(self.fit.shared_get()))
            self.fit.step_update(**adabelief_cfg)


def gather_epoch_blocks(local_flat, n_sources, group=None):
    """All-gather the per-epoch parameters so that every rank holds the full kwargs again."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(local_flat)
    world = dist.get_world_size(group)
    out = dict(local_flat)
    for k in ('a', 'dx', 'dy', 'alpha', 'mean'):
        parts = [None] * world
        dist.all_gather_object(parts, np.asarray(local_flat[k]), group=group)
        out[k] = np.concatenate(parts)
    return out
