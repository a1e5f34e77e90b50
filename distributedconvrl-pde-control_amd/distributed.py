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
    identical on every rank)."""

    def __init__(self, group=None, reduce_critic=True):
        self.group = group
        self.reduce_critic = bool(reduce_critic)
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self._views = {}

    def _view(self, model):
        key = int(model.handle.value)
        if key not in self._views:
            ptr, n = model.grad_buffer()
            ts = "<f8" if model.dtype == torch.float64 else "<f4"
            self._views[key] = torch.as_tensor(_DevArray(ptr, n, ts), device=model.device)
        return self._views[key]

    native = False        # torch.distributed issues the collective (host path of the process group)

    def all_reduce(self, model):
        if self.world_size == 1:
            return
        # through _Lib.note: a recorded control step (pipeline._eager) keeps the collective at its place between the
        # gradient launch and the ADAM launch and re-issues it on replay
        model.lib.note(self._reduce, self._view(model), model.stream)

    def _reduce(self, g, stream):
        if stream is not None:
            with torch.cuda.stream(stream):
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        return 0


class NativeGradReducer:
    """The same exchange through the library's own RCCL communicator (csrc/comm.hip: pdec_comm_create,
    pdec_allreduce_grads): ONE C call per all-reduce, enqueued on the network's (update) stream between the launch that
    leaves the flat gradient and the ADAM launch -- no torch.distributed host path (tensor wrappers, work objects, stream
    guards) on the critical stream, and, being a library call, recorded and replayed with the rest of a control step.
    The 128-byte ncclUniqueId travels once, at construction: over torch.distributed's object broadcast when a process
    group exists (any backend), or through `exchange` (a callable rank-0-bytes -> bytes, e.g. over a pipe)."""

    native = True

    def __init__(self, lib, rank=None, world_size=None, reduce_critic=True, exchange=None):
        import ctypes as C
        from . import _lib
        self.lib = lib
        self.reduce_critic = bool(reduce_critic)
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.rank, self.world_size = int(rank), int(world_size)
        buf = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(lib.pdec_comm_unique_id(C.cast(buf, C.c_void_p)))
        if self.world_size > 1:
            if exchange is not None:
                raw = exchange(bytes(buf.raw) if self.rank == 0 else None)
            else:
                box = [bytes(buf.raw) if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
            C.memmove(buf, raw, 128)
        self.comm = _lib.Handle()
        _lib.check(lib.pdec_comm_create(C.byref(self.comm), self.world_size, self.rank, C.cast(buf, C.c_void_p)))

    def all_reduce(self, model):
        if self.world_size == 1:
            return
        from . import _lib
        _lib.check(self.lib.pdec_allreduce_grads(self.comm, model.handle))

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
