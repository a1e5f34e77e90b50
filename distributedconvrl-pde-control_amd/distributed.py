"""Data-parallel pieces (the build's addition; the reference is single-process,
SURVEY.md §2.2/§8e): shard the batch of independent trajectories over one process per GPU and
sum-all-reduce the flattened actor / critic gradient once per update (RCCL over xGMI via
torch.distributed's "nccl" backend on GPUs, "gloo" in the CPU tests).  Nothing else crosses
GPUs; identical ADAM/Polyak steps keep the replicas bit-identical."""
import torch
import torch.distributed as dist


def shard_range(total, world_size, rank):
    """contiguous shard [lo, hi) of `total` trajectories for `rank` (remainder to the low ranks)"""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _DevArray:
    """__cuda_array_interface__ view of a library-owned device buffer"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class GradReducer:
    """all_reduce(model): sums the model's flat gradient buffer over the process group.
    reduce_critic=False is the north-star's "all-reduce of the policy gradient only": every rank trains its own
    critic on its shard of the trajectories (no critic exchange) and only the actor gradient is summed, so the
    ACTORS stay bit-identical replicas; reduce_critic=True also sums the critic gradient (all four networks
    identical on every rank).
    force_split=True keeps the data-parallel launch sequence (gradient pass -> reduce -> [all-reduce] -> ADAM/Polyak launch)
    with ONE rank as well: what `bench.py --split-update` times against the fused single-GPU finish."""

    def __init__(self, group=None, reduce_critic=True, force_split=False):
        self.group = group
        self.reduce_critic = bool(reduce_critic)
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force_split = bool(force_split)
        self._views = {}

    @property
    def active(self):
        """does policy.update take the split (data-parallel) sequence"""
        return self.world_size > 1 or self.force_split

    def _view(self, model):
        key = int(model.handle.value)
        if key not in self._views:
            ptr, n = model.grad_buffer()
            ts = "<f8" if model.dtype == torch.float64 else "<f4"
            self._views[key] = torch.as_tensor(_DevArray(ptr, n, ts), device=model.device)
        return self._views[key]

    native = False        # torch.distributed issues the collective (host path of the process group)

    def all_reduce(self, model, stream=None):
        """stream: the stream the collective is enqueued on (default: the network's own -- between the launch that leaves the
        gradient and the ADAM launch; pipeline.TrainPipeline passes a side stream and orders it with events)"""
        if self.world_size == 1:
            return
        # through _Lib.note: a recorded control step (pipeline._eager) keeps the collective at its place between the
        # gradient launch and the ADAM launch and re-issues it on replay
        model.lib.note(self._reduce, self._view(model), stream if stream is not None else model.stream)

    def _reduce(self, g, stream):
        if stream is not None:
            with torch.cuda.stream(stream):
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        return 0


class NativeGradReducer:
    """The same exchange through the library's own RCCL communicator (csrc/comm.hip: pdec_comm_create_timeout,
    pdec_allreduce_grads): ONE C call per all-reduce, enqueued on the network's (update) stream between the launch that
    leaves the flat gradient and the ADAM launch -- no torch.distributed host path (tensor wrappers, work objects, stream
    guards) on the critical stream, and, being a library call, recorded and replayed with the rest of a control step.

    Construction is a COLLECTIVE protocol in which every rank makes the same calls whatever fails:
      1. rank 0 creates the 128-byte ncclUniqueId and ALWAYS broadcasts (ok, id, error text) -- over torch.distributed's object
         broadcast when a process group exists (any backend), or through `exchange` (a callable `exchange(payload, op)`:
         op "broadcast" -- rank 0 passes the tuple, every rank gets it back, e.g. over a pipe; op "gather" -- every rank passes
         its own payload and gets the list of all ranks' payloads in rank order); if rank 0 failed, every rank raises here,
         before anyone enters the rendezvous;
      2. every rank enters RCCL's rendezvous with a bounded wait (`timeout_s`; a rank stuck without its peers comes back with
         an error instead of blocking for ever);
      3. the ranks all-gather whether they hold a communicator -- through the process group or `exchange(..., "gather")` --;
         unless all do, every rank drops its own and raises.
    So after the constructor either every rank has a working communicator or every rank got a PdecError.  (with an
    `exchange` that could only broadcast, step 3 was skipped and a peer of a failed rank kept a communicator whose first
    all-reduce would never return; such a callable is now refused before anything collective starts.)"""

    native = True

    def __init__(self, lib, rank=None, world_size=None, reduce_critic=True, exchange=None, timeout_s=90.0, force_split=False):
        import ctypes as C
        from . import _lib
        self.lib = lib
        self.comm = None
        self.reduce_critic = bool(reduce_critic)
        self.force_split = bool(force_split)
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.rank, self.world_size = int(rank), int(world_size)
        buf = (C.c_char * 128)()
        msg = (True, None, "")
        if self.rank == 0:
            rc = lib.pdec_comm_unique_id(C.cast(buf, C.c_void_p))
            msg = (rc == 0, bytes(buf.raw), "" if rc == 0 else lib.pdec_last_error().decode(errors="replace"))
        if self.world_size > 1 and exchange is not None:
            import inspect
            try:
                n_par = len([q for q in inspect.signature(exchange).parameters.values()
                             if q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD)])
                var = any(q.kind == q.VAR_POSITIONAL for q in inspect.signature(exchange).parameters.values())
            except (TypeError, ValueError):
                n_par, var = 2, False
            if n_par < 2 and not var:
                raise _lib.PdecError("NativeGradReducer: `exchange` must be exchange(payload, op) with op 'broadcast' and 'gather' -- "
                                     "without a gather the ranks cannot agree on whether every one of them holds a communicator")
        if self.world_size > 1:
            if exchange is not None:
                msg = exchange(msg if self.rank == 0 else None, "broadcast")
            else:
                box = [msg if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                msg = box[0]
        ok0, raw, err0 = msg
        if not ok0:
            raise _lib.PdecError(f"NativeGradReducer: rank 0 could not create the ncclUniqueId ({err0}); no rank entered the rendezvous")
        C.memmove(buf, raw, 128)
        comm = _lib.Handle()
        rc = lib.pdec_comm_create_timeout(C.byref(comm), self.world_size, self.rank, C.cast(buf, C.c_void_p),
                                          int(timeout_s * 1000) if self.world_size > 1 else 0)
        mine = rc == 0
        err = "" if mine else lib.pdec_last_error().decode(errors="replace")
        flags = [(mine, err)]
        if self.world_size > 1:
            if exchange is not None:
                flags = list(exchange((mine, err), "gather"))
                if len(flags) != self.world_size:
                    flags = flags + [(False, "exchange(..., 'gather') returned %d of %d verdicts" % (len(flags), self.world_size))]
            else:
                flags = [None] * self.world_size
                dist.all_gather_object(flags, (mine, err))
        if not all(f[0] for f in flags):
            if mine:
                lib.pdec_destroy(comm)
            bad = [(r, f[1]) for r, f in enumerate(flags) if not f[0]]
            raise _lib.PdecError("NativeGradReducer: pdec_comm_create_timeout (ncclCommInitRank) failed on rank(s) "
                                 + "; ".join(f"{r}: {e}" for r, e in bad) + (f" -- own: {err}" if err else ""))
        self.comm = comm

    @property
    def active(self):
        return self.world_size > 1 or self.force_split

    def all_reduce(self, model, stream=None):
        if not self.active:
            return
        import ctypes as C
        from . import _lib
        if stream is None:
            _lib.check(self.lib.pdec_allreduce_grads(self.comm, model.handle))
        else:
            _lib.check(self.lib.pdec_allreduce_grads_on(self.comm, model.handle, C.c_void_p(stream.cuda_stream)))

    def verify_against_torch(self, device, n=20441, timeout_s=30.0, group=None):
        """one all-reduce of a KNOWN buffer through this communicator and one through torch.distributed, before anything is
        timed: both must equal the closed-form sum bit for bit (integer-valued floats: exact in any reduction order).  The
        native call is watched with a deadline (a collective that never completes returns False instead of hanging the
        caller).  Every rank makes the same calls; returns this rank's verdict -- MIN-reduce it over the ranks."""
        import time
        from . import _lib
        i = torch.arange(n, device=device, dtype=torch.float32)
        base = (i % 251) - 125.0
        x = base * float(self.rank + 1)
        y = x.clone()
        want = base * float(self.world_size * (self.world_size + 1) // 2)
        st = torch.cuda.Stream(device=device)
        st.wait_stream(torch.cuda.current_stream(device))
        rc = self.lib.pdec_allreduce(self.comm, _lib.ptr(x), n, _lib.PDEC_F32, st.cuda_stream)
        ok = rc == 0
        t0 = time.perf_counter()
        while ok and not st.query():
            if time.perf_counter() - t0 > timeout_s:
                ok = False
                break
            time.sleep(0.002)
        if self.world_size > 1 and dist.is_initialized():
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        else:
            y = want.clone()
        if ok:
            torch.cuda.synchronize(device)
            ok = bool(torch.equal(x, want)) and bool(torch.equal(y, want))
        return ok

    def close(self):
        if self.comm is not None:
            self.lib.pdec_destroy(self.comm)
            self.comm = None


def all_reduce_host_grads(grads, group=None):
    """CPU/gloo form used by the world_size-2 tests: list of numpy arrays summed in place"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return grads
    flat = torch.cat([torch.as_tensor(g).reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    out, o = [], 0
    for g in grads:
        out.append(flat[o:o + g.size].reshape(g.shape).numpy().copy())
        o += g.size
    return out
