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


_HOST_GROUPS = {}


def host_group(group=None):
    """A gloo group over the ranks of ``group`` for the object collectives of this module (errors, IPC handles, per-epoch
    blocks): they carry pickled host objects, which an NCCL group would stage through device tensors on whatever device
    happens to be current.  ``group`` itself when it is gloo already; otherwise a gloo group created once per group -
    COLLECTIVELY: every rank of ``group`` must make its first call at the same point (PeerGroup and ShardedJointOptimizer do
    it in their constructors)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_backend(group) == 'gloo':
        return group
    key = id(group) if group is not None else None
    if key not in _HOST_GROUPS:
        ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
        _HOST_GROUPS[key] = dist.new_group(ranks=ranks, backend='gloo')
    return _HOST_GROUPS[key]


def raise_together(err, what, group=None):
    """Every rank hands in its error (None = fine); if any rank has one, EVERY rank raises - a rank that raised alone would
    leave the others waiting for it in their next collective until the backend's time-out."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if err:
            raise RuntimeError(f'{what}: {err}')
        return
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, err, group=host_group(group))
    bad = [(r, e) for r, e in enumerate(parts) if e]
    if bad:
        raise RuntimeError(f'{what}: ' + '; '.join(f'rank {r}: {e}' for r, e in bad))


class PeerGroup:
    """One-shot peer-memory all-reduce of a fit's shared block (include/lcmi.h "peer group", csrc/peer.hip): every rank
    publishes its block in an exchange buffer the other ranks map with HIP IPC and reads the N - 1 peers directly, adding
    in rank order.  The IPC handles travel over ``group`` (any torch.distributed group of the same ranks, e.g. gloo)."""

    def __init__(self, local_fit, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        self._l = _lib.lib()
        self.fit = local_fit
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.group = group = host_group(group)
        _, count = local_fit.shared_buffer()
        # a rank whose step fails still takes part in the exchanges that follow, so that every rank raises instead of one
        # raising and the others waiting for it in a collective
        self.h, err, raw = None, None, b''
        try:
            h = C.c_void_p()
            local_fit.ctx.check(self._l.lc_peer_group_create(local_fit.ctx.h, count, self.rank, self.world, C.byref(h)), 'lc_peer_group_create')
            self.h = h
            mine = C.create_string_buffer(_lib.IPC_HANDLE_BYTES)
            local_fit.ctx.check(self._l.lc_peer_group_export(self.h, mine, _lib.IPC_HANDLE_BYTES), 'lc_peer_group_export')
            raw = bytes(mine.raw)
        except Exception as e:
            err = repr(e)
        parts = [None] * self.world
        dist.all_gather_object(parts, (err, raw), group=group)
        self._raise_if_any([p[0] for p in parts], 'exchange buffer')
        err = None
        try:
            allh = C.create_string_buffer(b''.join(p[1] for p in parts), _lib.IPC_HANDLE_BYTES * self.world)
            local_fit.ctx.check(self._l.lc_peer_group_connect(self.h, allh, _lib.IPC_HANDLE_BYTES), 'lc_peer_group_connect')
        except Exception as e:
            err = repr(e)
        parts = [None] * self.world
        dist.all_gather_object(parts, err, group=group)   # also the barrier: every rank has mapped every peer before anyone publishes
        self._raise_if_any(parts, 'mapping the peers')

    def _raise_if_any(self, errors, what):
        bad = [(r, e) for r, e in enumerate(errors) if e]
        if bad:
            self.close(barrier=False)   # (every rank is here with the same list, and no kernel has touched the buffers yet)
            raise RuntimeError(f'peer group ({what}): ' + '; '.join(f'rank {r}: {e}' for r, e in bad))

    def callback(self):
        """(function pointer, user pointer) for lc_joint_run_sharded: the library's own lc_peer_allreduce - no Python in the loop."""
        import ctypes as C
        return C.cast(self._l.lc_peer_allreduce, C.c_void_p), self.h

    def check(self):
        """Raises on THIS rank if one of its waits ran out (synchronises the library's stream)."""
        self.fit.ctx.check(self._l.lc_peer_group_status(self.h), 'peer all-reduce')

    def status(self):
        """The same as a string (None = every wait was answered), for raise_together."""
        try:
            self.check()
            return None
        except Exception as e:  # noqa: BLE001
            return repr(e)

    def check_together(self):
        """Every rank raises if ANY rank's wait ran out (one small host collective)."""
        raise_together(self.status(), 'peer all-reduce', self.group)

    def close(self, barrier=True):
        """Unmaps the peers and frees the exchange buffer.  A peer may still be inside its last read of this rank's buffer
        when the local stream has drained, so the ranks first meet at a host barrier (bounded: a rank that died must not
        hang the others' clean-up; after a time-out the buffer is freed anyway - the peer is gone or broken)."""
        if getattr(self, 'h', None):
            if barrier:
                try:
                    import datetime
                    import torch.distributed as dist
                    if dist.is_initialized() and self.world > 1:
                        if hasattr(self.fit.ctx, 'synchronize'):
                            self.fit.ctx.synchronize()
                        dist.monitored_barrier(group=self.group, timeout=datetime.timedelta(seconds=20))
                except Exception:  # noqa: BLE001
                    pass
            self._l.lc_peer_group_destroy(self.h)
            self.h = None

    def __del__(self):
        # (not while the interpreter shuts down: objects are then torn down in no particular order, and destroying a device
        #  object whose context has already gone is a crash at exit; the process is about to return everything anyway)
        try:
            import sys
            if not sys.is_finalizing():
                self.close(barrier=False)   # (a collective has no place in a finaliser: the caller's close() is the clean path)
        except Exception:
            pass


class RcclGroup:
    """RCCL communicator owned by the library (include/lcmi.h "RCCL group", csrc/rccl.hip): librccl is loaded at run time,
    the unique id of rank 0 travels over ``group`` (any torch.distributed group of the same ranks - gloo will do: the
    communicator is the library's own, not torch's), and ``callback()`` hands lc_joint_run_sharded the C entry point
    lc_rccl_allreduce - ncclAllReduce in place on the library's stream, no Python inside the loop.  One rank per GPU (RCCL
    refuses two ranks on one device)."""

    def __init__(self, ctx, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        self._l = _lib.lib()
        self.ctx = ctx
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.group = group = host_group(group)
        self.h, err, raw = None, None, b''
        if self.rank == 0:
            try:
                buf = C.create_string_buffer(_lib.RCCL_ID_BYTES)
                rc = self._l.lc_rccl_unique_id(buf, _lib.RCCL_ID_BYTES)
                if rc:
                    raise RuntimeError(f'lc_rccl_unique_id failed ({rc}): librccl not loadable?')
                raw = bytes(buf.raw)
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        parts = [None] * self.world
        dist.all_gather_object(parts, (err, raw), group=group)
        if parts[0][0]:
            raise RuntimeError(f'RCCL group (unique id): rank 0: {parts[0][0]}')
        err = None
        try:
            h = C.c_void_p()
            uid = C.create_string_buffer(parts[0][1], _lib.RCCL_ID_BYTES)
            ctx.check(self._l.lc_rccl_group_create(ctx.h, uid, _lib.RCCL_ID_BYTES, self.rank, self.world, C.byref(h)),
                      'lc_rccl_group_create')
            self.h = h
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        raise_together(err, 'RCCL group (ncclCommInitRank)', group)

    def callback(self):
        """(function pointer, user pointer) for lc_joint_run_sharded: the library's own lc_rccl_allreduce."""
        import ctypes as C
        return C.cast(self._l.lc_rccl_allreduce, C.c_void_p), self.h

    def all_reduce(self, dev_ptr, count, stream_ptr):
        """One all-reduce of ``count`` floats at ``dev_ptr``, enqueued on ``stream_ptr`` (what the loop calls per iteration)."""
        import ctypes as C
        self.ctx.check(self._l.lc_rccl_allreduce(self.h, C.c_void_p(dev_ptr), int(count), C.c_void_p(stream_ptr)), 'lc_rccl_allreduce')

    @property
    def calls(self):
        import ctypes as C
        n = C.c_longlong()
        self._l.lc_rccl_group_info(self.h, None, None, C.byref(n))
        return n.value

    def close(self):
        if getattr(self, 'h', None):
            self._l.lc_rccl_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            import sys
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass


class ShardedJointOptimizer:
    """Drives ``n_iter`` AdaBelief iterations of a sharded joint fit.

    ``local_fit`` is the rank's device object (``lightcurver_amd.joint.JointFit`` over the local epochs) or
    anything with the same four methods: step_local(), shared_get() -> float array, shared_set(array),
    step_update(**adabelief_cfg).
    """

    def __init__(self, local_fit, group=None, peer=None, rccl=None):
        """peer: a ``PeerGroup`` over the same ranks - the shared block is then reduced by the one-shot peer-memory kernel
        instead of the process group's collective.  rccl: an ``RcclGroup`` over the same ranks - reduced by the library's own
        RCCL communicator, called from the C++ loop (no Python per iteration); ``group`` then only carries host objects."""
        self.fit = local_fit
        self.group = group
        self.peer = peer
        self.rccl = rccl
        self.host_group = host_group(group)   # (collective on first use: see host_group)
        self._dev = None  # (tensor view of the shared block, torch ExternalStream of the library's stream)
        self._ref_agreed = False

    @property
    def transport(self):
        """'peer' (one-shot peer-memory kernel), 'rccl' (in place in device memory) or 'gloo' (staged through the host)."""
        if self.peer is not None:
            return 'peer'
        if self.rccl is not None:
            return 'rccl-native'
        return 'rccl' if self._device_collective() else 'gloo'

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
        if hasattr(self.fit, 'run_sharded'):
            # the loop itself runs in C++ (lc_joint_run_sharded): per iteration step_local, this callback, step_update
            if self.peer is not None:
                # a wait that ran out leaves that rank's block un-reduced and is latched in its status word only: the ranks
                # agree on the outcome, so that all of them raise instead of one raising and the rest walking into their
                # next collective alone
                err = None
                try:
                    fn, user = self.peer.callback()
                    self.fit.run_sharded(int(n_iter), fn, user, **adabelief_cfg)
                    self.peer.check()
                except Exception as e:  # noqa: BLE001
                    err = repr(e)
                raise_together(err, 'sharded joint fit (peer all-reduce)', self.host_group)
                return
            if self.rccl is not None:
                # the library's own communicator: lc_rccl_allreduce is the callback, the loop never leaves C++
                fn, user = self.rccl.callback()
                self.fit.run_sharded(int(n_iter), fn, user, **adabelief_cfg)
                return
            failure = []

            def reduce_block(_user, _buf, _count, _stream):
                try:
                    if on_device:
                        self.all_reduce_device()       # RCCL, in place, ordered on the library's stream
                    else:
                        self.fit.shared_set(self.all_reduce(self.fit.shared_get()))
                    return 0
                except BaseException as exc:  # noqa: BLE001  (ctypes would swallow it)
                    failure.append(exc)
                    return 1

            from . import _lib
            cb = _lib.ALLREDUCE_FN(reduce_block)
            try:
                self.fit.run_sharded(int(n_iter), cb, None, **adabelief_cfg)
            finally:
                if failure:
                    raise failure[0]
            return
        for _ in range(int(n_iter)):   # (objects without the entry point: the protocol stand-ins of the CPU tests)
            self.fit.step_local()
            if on_device:
                self.all_reduce_device()
            else:
                self.fit.shared_set(self.all_reduce(self.fit.shared_get()))
            self.fit.step_update(**adabelief_cfg)


PER_EPOCH = ('a', 'dx', 'dy', 'mean')     # blocks whose entries belong to one epoch (a: M per epoch)


def sharded_lbfgs(optimizer, free, maxiter, lower=None, upper=None):
    """The L-BFGS-B stage of the two-stage fit (reference lightcurver/processes/roi_modelling.py:278-280: positions, fluxes
    and shifts with the background held) with the epochs spread over the ranks of ``optimizer`` (a ShardedJointOptimizer).

    Every rank runs the same scipy L-BFGS-B on the FULL vector [shared blocks | per-epoch blocks of all epochs]; an evaluation
    is local forward/backward (step_local), the all-reduce of the shared block the AdaBelief loop also uses, and
    lc_joint_step_grad: the loss of the whole fit and the gradients of the shared parameters come out complete and identical
    on every rank, the gradients of the per-epoch parameters for the local epochs - those are all-gathered.  Same numbers in,
    same iterates on all ranks.  Returns (loss history per accepted iterate, scipy result); the fit holds the final point.

    lower / upper: dicts name -> scalar or array over the rank's LOCAL entries of that block (missing = unbounded)."""
    import torch
    import torch.distributed as dist
    from scipy.optimize import minimize
    fit, group = optimizer.fit, optimizer.group
    hgroup = getattr(optimizer, 'host_group', None) or host_group(group)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    free = [k for k in ('c_x', 'c_y', 'h', 'a', 'dx', 'dy', 'mean') if k in free]
    fit.set_free(free)
    optimizer._agree_flux_reference()
    # (set_params(a=...) re-centres the flux moments on the LOCAL mean flux: every evaluation puts the agreed reference back)
    ref = fit.get_flux_reference() if hasattr(fit, 'get_flux_reference') else None
    M = fit.M
    cur = {k: np.asarray(v, np.float64) for k, v in fit.get_params().items()}

    def gather(vec):
        """local per-epoch block -> the block of all epochs, in rank order"""
        if world == 1:
            return np.asarray(vec, np.float64)
        parts = [None] * world
        dist.all_gather_object(parts, np.asarray(vec, np.float64), group=hgroup)
        return np.concatenate(parts)

    def gather_all(vecs, err):
        """the local per-epoch blocks of one evaluation and this rank's error in ONE host collective: every rank learns of a
        failure anywhere (and raises with it) before it uses the numbers"""
        if world == 1:
            if err:
                raise RuntimeError(f'sharded L-BFGS-B: {err}')
            return [np.asarray(v, np.float64) for v in vecs]
        parts = [None] * world
        dist.all_gather_object(parts, (err, [np.asarray(v, np.float64) for v in vecs]), group=hgroup)
        bad = [(r, p[0]) for r, p in enumerate(parts) if p[0]]
        if bad:
            raise RuntimeError('sharded L-BFGS-B: ' + '; '.join(f'rank {r}: {e}' for r, e in bad))
        return [np.concatenate([p[1][i] for p in parts]) for i in range(len(vecs))]

    sizes_local = {k: cur[k].size for k in free}
    full = {k: (gather(cur[k]) if k in PER_EPOCH else cur[k]) for k in free}
    offs, n = {}, 0
    for k in free:
        offs[k] = (n, n + full[k].size)
        n += full[k].size
    # where the rank's own entries sit inside the gathered per-epoch blocks
    counts = [None] * world
    if world > 1:
        dist.all_gather_object(counts, int(fit.E), group=hgroup)
    else:
        counts = [int(fit.E)]
    rank = dist.get_rank(group) if world > 1 else 0
    e0 = int(sum(counts[:rank]))

    def own(k):
        per = M if k == 'a' else 1
        return slice(e0 * per, (e0 + fit.E) * per)

    def bound(table, fill):
        out = np.full(n, fill)
        for k in free:
            if table and k in table:
                loc = np.broadcast_to(np.asarray(table[k], np.float64), (sizes_local[k],))
                out[offs[k][0]:offs[k][1]] = gather(loc) if k in PER_EPOCH else loc
        return out

    lo, hi = bound(lower, -np.inf), bound(upper, np.inf)
    x0 = np.concatenate([full[k] for k in free]) if free else np.zeros(0)
    hist, last = [], {}

    def fun(x):
        p = {}
        for k in free:
            blk = x[offs[k][0]:offs[k][1]]
            p[k] = blk[own(k)] if k in PER_EPOCH else blk
        # A failure on one rank must stop every rank at the same evaluation: none may optimise on with local-only numbers or
        # wait alone in a collective.  The local step's outcome is agreed BEFORE the data all-reduce (a rank that skipped it
        # would leave the others waiting in it); what the all-reduce itself reports - a peer wait that ran out leaves that
        # rank's block un-reduced and latches a status word - travels with the gradient gather below.
        err, loss, g = None, float('nan'), None
        try:
            fit.set_params(**p)
            if ref is not None:
                fit.set_flux_reference(ref)
            fit.step_local()
        except Exception as e:  # noqa: BLE001
            if world == 1:
                raise
            err = repr(e)
        if world > 1:
            raise_together(err, 'sharded L-BFGS-B (local step)', hgroup)
        try:
            if optimizer.peer is not None:
                import ctypes as C
                fn, user = optimizer.peer.callback()
                ptr, count = fit.shared_buffer()
                stream, _ = fit.ctx.stream()
                rc = optimizer.peer._l.lc_peer_allreduce(user, C.c_void_p(ptr), count, C.c_void_p(stream))
                if rc:
                    raise RuntimeError(f'lc_peer_allreduce failed ({rc})')
                optimizer.peer.check()       # (synchronises the stream, as the gradient read-back below would anyway)
            elif getattr(optimizer, 'rccl', None) is not None:
                ptr, count = fit.shared_buffer()
                optimizer.rccl.all_reduce(ptr, count, fit.ctx.stream()[0])
            elif optimizer._device_collective():
                optimizer.all_reduce_device()
            else:
                fit.shared_set(optimizer.all_reduce(fit.shared_get()))
            loss, g = fit.step_grad(tuple(free))
        except Exception as e:  # noqa: BLE001
            if world == 1:
                raise
            err = repr(e)
        per_epoch = [k for k in free if k in PER_EPOCH]
        local = [g[k] if g is not None else np.zeros(sizes_local[k]) for k in per_epoch]
        gathered = dict(zip(per_epoch, gather_all(local, err)))
        grad = np.concatenate([gathered[k] if k in PER_EPOCH else np.asarray(g[k], np.float64) for k in free])
        last['x'], last['val'] = np.array(x, copy=True), float(loss)
        return float(loss), grad

    def record(xk):
        hist.append(last['val'] if 'x' in last and np.array_equal(last['x'], xk) else fun(xk)[0])

    res = minimize(fun, x0, jac=True, method='L-BFGS-B', bounds=list(zip(lo, hi)), callback=record,
                   options=dict(maxiter=int(maxiter)))
    fun(res.x)   # the fit holds the final point (and its loss is the last evaluation)
    return np.asarray(hist, np.float64), res


ShardedJointOptimizer.run_lbfgs = lambda self, free, maxiter, lower=None, upper=None: sharded_lbfgs(self, free, maxiter, lower, upper)
ShardedJointOptimizer.run_lbfgs.__doc__ = 'The L-BFGS-B stage on the sharded fit: see sharded_lbfgs.'


def gather_epoch_blocks(local_flat, n_sources, group=None):
    """All-gather the per-epoch parameters so that every rank holds the full kwargs again."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(local_flat)
    world = dist.get_world_size(group)
    out = dict(local_flat)
    keys = ('a', 'dx', 'dy', 'alpha', 'mean')
    parts = [None] * world
    dist.all_gather_object(parts, [np.asarray(local_flat[k]) for k in keys], group=host_group(group))
    for i, k in enumerate(keys):
        out[k] = np.concatenate([p[i] for p in parts])
    return out
