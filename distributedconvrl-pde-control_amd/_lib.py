"""ctypes binding of libpdeconv.so (the C ABI declared in include/pdeconv.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails,
an exception is raised.  Nothing here imports oracle/."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDEC_LIB_PATH") or os.path.join(_HERE, "libpdeconv.so")      # (override: A/B builds, diagnostics)

PDEC_F32, PDEC_F64 = 0, 1
PDE_KS_CNAB2, PDE_KSEG_RK4, PDE_KS_RK4_FD, PDE_FLUID_RK4, PDE_KSEG2D_RK4 = 0, 1, 2, 3, 4
ACT_IDENTITY, ACT_RELU, ACT_TANH = 0, 1, 2

Handle = C.c_uint64


class PdecError(RuntimeError):
    pass


class EnvCfg(C.Structure):
    """mirror of `struct pdec_env_cfg` (include/pdeconv.h)"""
    _fields_ = [
        ("pde_kind", C.c_int), ("dtype", C.c_int), ("B", C.c_int), ("N", C.c_int),
        ("n_species", C.c_int), ("S", C.c_int), ("A", C.c_int), ("window", C.c_int),
        ("temporal_steps", C.c_int), ("mono", C.c_int), ("K", C.c_int), ("check_max_value", C.c_int),
        ("Lx", C.c_double), ("dt", C.c_double), ("mu", C.c_double), ("max_value", C.c_double),
        ("sensor_scale", C.c_double), ("agent_power", C.c_double), ("reward_in_scale", C.c_double),
        ("reward_offset", C.c_double), ("reward_power", C.c_double), ("reward_denom", C.c_double),
        ("action_punish", C.c_double), ("delta_action_punish", C.c_double),
        ("ifpad", C.c_int), ("sensors_per_axis", C.c_int), ("nu", C.c_double),
        ("Ny", C.c_int), ("integrator", C.c_int), ("memory_size", C.c_int),
    ]


_vp, _i, _d, _sz, _u64, _i64 = C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_uint64, C.c_int64
_pi32 = C.POINTER(C.c_int32)
_pd = C.POINTER(C.c_double)

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "pdec_init": [_i], "pdec_shutdown": [], "pdec_version": [], "pdec_device_count": [C.POINTER(_i)],
    "pdec_malloc": [C.POINTER(_vp), _sz], "pdec_free": [_vp],
    "pdec_memcpy_h2d": [_vp, _vp, _sz], "pdec_memcpy_d2h": [_vp, _vp, _sz], "pdec_memset": [_vp, _i, _sz],
    "pdec_set_stream": [Handle, _vp], "pdec_sync": [Handle], "pdec_destroy": [Handle],
    "pdec_stream_create": [C.POINTER(_vp), _i], "pdec_stream_destroy": [_vp],
    "pdec_env_part_streams": [Handle, C.POINTER(_i)], "pdec_env_set_part_streams": [Handle, C.POINTER(_vp), _i],
    "pdec_prof_enable": [Handle, _i], "pdec_prof_reset": [Handle],
    "pdec_prof_get": [Handle, C.c_char_p, _pd, C.POINTER(_i)],
    "pdec_env_create": [C.POINTER(Handle), C.POINTER(EnvCfg), _pd, _pd, _pi32],
    "pdec_fluid_env_create": [C.POINTER(Handle), C.POINTER(EnvCfg), _i, _i, _pd, _pi32, _pd, _pi32, _pi32],
    "pdec_kseg2d_env_create": [C.POINTER(Handle), C.POINTER(EnvCfg), _i, _i, _i, _pi32, _pi32, _i, _pi32],
    "pdec_fluid_ic": [Handle, _pd, _i, _vp],
    "pdec_actuate": [Handle, _vp, _vp],
    "pdec_pde_step": [Handle, _vp, _vp, _vp, _vp],
    "pdec_featurize": [Handle, _vp, _vp, _vp], "pdec_featurize_action": [Handle, _vp, _vp, _vp, _vp], "pdec_mlp_set_noise_rows": [Handle, _i],
    "pdec_reward": [Handle, _vp, _vp, _vp, _vp],
    "pdec_env_step": [Handle] + [_vp] * 9,
    "pdec_rhs_eval": [Handle, _vp, _vp, _vp],
    "pdec_env_set_terminal_out": [Handle, _vp],
    "pdec_pde_step_host": [Handle, _vp, _vp, _vp, _vp],
    "pdec_env_step_host": [Handle] + [_vp] * 9,
    "pdec_mlp_create": [C.POINTER(Handle), _i, _i, _pi32, _pi32, _vp, _i],
    "pdec_mlp_num_params": [Handle, C.POINTER(_i)],
    "pdec_mlp_set_params": [Handle, _vp], "pdec_mlp_get_params": [Handle, _vp],
    "pdec_mlp_copy": [Handle, Handle],
    "pdec_mlp_forward": [Handle, _vp, _i, _vp],
    "pdec_mlp_backward": [Handle, _vp, _vp, _i, _vp, _vp],
    "pdec_mlp_grad_buffer": [Handle, C.POINTER(_vp), C.POINTER(_i)],
    "pdec_adam_step": [Handle, _d, _d, _d, _d],
    "pdec_adam_get_state": [Handle, _vp, _vp, _pd], "pdec_adam_set_state": [Handle, _vp, _vp, _pd],
    "pdec_polyak": [Handle, Handle, _d],
    "pdec_adam_polyak_step": [Handle, Handle, _d, _d, _d, _d, _d],
    "pdec_policy_act_rng": [Handle, _vp, _i, _d, _d, _i, _u64, _u64, _vp],
    "pdec_policy_act_rng_as": [Handle, _i, _vp, _i, _d, _d, _i, _u64, _u64, _vp, C.POINTER(_i)],
    "pdec_step_glue_served": [Handle, Handle, _i, _i, _i, _i64, C.POINTER(_i)],
    "pdec_set_launch_sync": [Handle, _vp, _i64, _vp, _i64],
    "pdec_launch_sync_timeouts": [C.POINTER(_i)],
    "pdec_streams_run_side_by_side": [_vp, _vp, C.POINTER(_i)],
    "pdec_step_glue": [Handle, Handle, _i, _vp, _vp, _i, _i, _vp, _vp, _i64, _i64, _i64, _i, _vp, _i, _d, _d, _u64, _u64, _vp, _vp,
                       _vp, _i64, _i64, _i64, Handle, C.POINTER(_i)],
    "pdec_ddpg_update_async": [Handle] * 4 + [_vp] * 5 + [_i, _d, _d, _i, _d, _d, _vp],
    "pdec_ddpg_update_small": [Handle] * 4 + [_vp] * 7 + [_i, _i, _d, _d, _i, _d, _d, _vp],
    "pdec_ddpg_update_critic_async": [Handle] * 4 + [_vp] * 5 + [_i, _d, _d, _i, _d, _vp],
    "pdec_ddpg_update_actor_async": [Handle] * 4 + [_vp, _i, _d, _d, _vp],
    "pdec_policy_act": [Handle, _vp, _vp, _i, _d, _d, _vp],
    "pdec_rollout": [Handle, Handle, _i, _vp, _vp, _vp, _d, _d, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdec_randn": [Handle, _vp, _sz, _i, _u64, _u64],
    "pdec_policy_act_rng_dev": [Handle, _vp, _i, _d, _d, _i, _u64, _vp],
    "pdec_mlp_acts_on_published_copy": [Handle, C.POINTER(_i)],
    "pdec_noise_counter_set": [Handle, _u64], "pdec_noise_counter_get": [Handle, C.POINTER(_u64)],
    "pdec_reward_mean": [Handle, _vp, _i, _vp], "pdec_ddpg_set_reward_mean": [Handle, _vp],
    "pdec_env_set_reward_partials_out": [Handle, _vp, C.POINTER(_i)], "pdec_ddpg_set_reward_partials": [Handle, _vp, _i],
    "pdec_ddpg_update_small_rng": [Handle] * 4 + [_vp] * 4 + [_i, _i, _u64, _u64, _i64, _i64, _i64, _i, _d, _d, _i, _d, _d, _vp],
    "pdec_replay_push_sa": [Handle, _vp, _vp, _i64, _i, _i, _i64, _vp, _vp, _i64, _i],
    "pdec_replay_push_rt": [Handle, _vp, _vp, _i64, _i64, _vp, _vp, _i, _i, _i64, _i],
    "pdec_replay_sample": [Handle, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _i64, _i64, _u64, _u64, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdec_set_episode_halt": [Handle, _vp],
    "pdec_env_autoreset": [Handle] + [_vp] * 8,
    "pdec_env_random_init": [Handle, _u64, _u64, _vp],
    "pdec_capture_begin": [Handle], "pdec_capture_end": [Handle, C.POINTER(Handle)],
    "pdec_graph_launch": [Handle, _vp], "pdec_graph_num_nodes": [Handle, C.POINTER(_i)],
    "pdec_event_create": [C.POINTER(Handle)], "pdec_event_record": [Handle, _vp], "pdec_stream_wait_event": [_vp, Handle], "pdec_mlp_set_stop_event": [Handle, Handle], "pdec_mlp_flush_stop_event": [Handle],
    "pdec_mlp_set_reduce_event": [Handle, Handle],
    "pdec_env_set_simd_sharing": [Handle, C.c_int, C.POINTER(C.c_int)],
    "pdec_ddpg_critic_grads": [Handle] * 4 + [_vp] * 5 + [_i, _d, _i, _d, _vp],
    "pdec_ddpg_actor_grads": [Handle, Handle, _vp, _i, _d, _vp],
    "pdec_ddpg_update": [Handle] * 4 + [_vp] * 5 + [_i, _d, _d, _i, _d, _d, _pd, _pd],
    "pdec_comm_unique_id": [_vp], "pdec_comm_create": [C.POINTER(Handle), _i, _i, _vp],
    "pdec_comm_create_timeout": [C.POINTER(Handle), _i, _i, _vp, _i],
    "pdec_allreduce_grads": [Handle, Handle], "pdec_allreduce": [Handle, _vp, _sz, _i, _vp],
    "pdec_allreduce_grads_on": [Handle, Handle, _vp],
}
# include/pdeconv_debug.h: unit-test / measurement entry points (tests/, bench.py, tools/), not part of the drop-in surface
DEBUG_SIGNATURES = {
    "pdec_debug_wave_fft": [_vp, _vp, _i, _i, _i],
    "pdec_debug_critic_stamps": [Handle, _i, _pd],
    "pdec_debug_kseg2d_probe": [Handle, _i, _i, _i, _pd],
    "pdec_debug_spin_us": [_vp, _d],
}
_RESTYPES = {"pdec_last_error": C.c_char_p}

_lib = None


class _Lib:
    """The loaded library.  Attribute access returns the C functions; while `record_into(list)` is active every call is
    also appended to the list as (function, args) -- pipeline.py replays such a list to re-issue a control step whose
    arguments are those of an earlier one without going through the Python layers again."""

    def __init__(self, cdll):
        self._c = cdll
        self._rec = None

    # never part of a recorded control step: object teardown reaches the library from __del__ of unrelated Python objects
    # whenever the garbage collector runs -- also in the middle of a recording -- and a replayed pdec_destroy names a dead handle
    _NEVER_RECORDED = frozenset({"pdec_destroy", "pdec_free", "pdec_shutdown", "pdec_stream_create", "pdec_stream_destroy"})

    def __getattr__(self, name):
        f = getattr(self._c, name)
        if name in self._NEVER_RECORDED:
            setattr(self, name, f)
            return f

        def call(*a, _f=f):
            if self._rec is not None:
                self._rec.append((_f, a))
            return _f(*a)
        call.__name__ = name
        setattr(self, name, call)
        return call

    def record_into(self, calls):
        self._rec = calls

    def note(self, fn, *args):
        """run a call that must be re-EVALUATED at replay -- a non-library call (a torch.distributed all-reduce between two
        library launches) or a host function whose library calls take step-dependent scalars (the replay pushes / the sample
        of the device replay: ring positions, counters) -- and, while a recording is active, keep it in the list at its
        place.  `fn` must return None (or 0)."""
        if self._rec is not None:
            self._rec.append((fn, args))
            rec, self._rec = self._rec, None      # the library calls `fn` makes itself are part of `fn`, not of the list
            try:
                return fn(*args)
            finally:
                self._rec = rec
        return fn(*args)


def load():
    """Load libpdeconv.so (once).  Raises PdecError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PdecError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C distributedconvrl-pde-control_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, args in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.pdec_last_error.argtypes = []
    lib.pdec_last_error.restype = C.c_char_p
    _lib = _Lib(lib)
    return _lib


def check(rc):
    if rc != 0:
        raise PdecError(f"libpdeconv error {rc}: {load().pdec_last_error().decode(errors='replace')}")


_inited = set()


def init(device=0):
    """Select `device` for this process.  The design is ONE PROCESS PER GPU: handles carry no device ordinal and every
    hipMalloc / launch of the library goes to the device selected here, so a second, different ordinal in the same
    process is refused instead of silently running one GPU's handles on another."""
    lib = load()
    device = int(device)
    if device not in _inited:
        if _inited:
            raise PdecError(f"libpdeconv is already bound to device {sorted(_inited)[0]} in this process; device {device} "
                            "needs its own process (one process per GPU, e.g. torch.distributed.run / bench.py --gpus N)")
        check(lib.pdec_init(device))
        _inited.add(device)
    return lib


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float32:
        return PDEC_F32
    if torch_dtype == torch.float64:
        return PDEC_F64
    raise PdecError(f"unsupported dtype {torch_dtype}")


def ptr(t):
    """device (or host) address of a torch tensor / numpy array / None"""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        assert t.is_contiguous(), "libpdeconv needs contiguous tensors"
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


_streams = {}   # hipStream_t address -> torch.cuda.ExternalStream, kept for the life of the process (see make_stream)


def make_stream(level=0, device=0):
    """A non-blocking HIP stream of the library at priority LEVEL (-1 high, 0 normal, +1 low), wrapped as a
    torch.cuda.ExternalStream.  Unlike torch.cuda.Stream(priority=...), which hands out the next of a 32-stream pool per
    level -- whichever hardware queue that position happens to sit on --, every call makes a NEW stream, so two pipeline
    streams at two different levels never share a hardware queue (include/pdeconv.h, pdec_stream_create).  The stream
    lives until destroy_stream() or the end of the process: library objects hold its raw address (pdec_set_stream), so
    it is never released behind their back by garbage collection."""
    import torch
    lib = init(device)
    out = _vp()
    # pdec_stream_create makes the stream on the CURRENT HIP device: make `device` current for the call, so that the wrapper
    # below names the device the stream really lives on
    with torch.cuda.device(int(device)):
        check(lib.pdec_stream_create(C.byref(out), int(level)))
    s = torch.cuda.ExternalStream(out.value, device=torch.device("cuda", int(device)))
    _streams[out.value] = s
    return s


def make_streams(levels, device=0):
    """make_stream for each level, BACK TO BACK: hardware queues sit on the GPU's four compute pipes in the order they are
    made, and two busy queues on one pipe take turns -- so the (at most four) streams that work at the same time are made in
    one go: e.g. `s_env, s_upd, *parts = make_streams((-1, 0, -1, -1))` for a pipeline whose environment runs three parts (the part
    streams at the env stream's level: part 0 runs on the env stream itself, and parts of one level advance evenly)."""
    levels = list(levels)
    if len(levels) > 4:
        raise PdecError("make_streams: more than four streams cannot sit on four different compute pipes")
    return [make_stream(lv, device) for lv in levels]


def destroy_stream(s):
    """Releases a stream made by make_stream (the caller guarantees that no library object is still set to it)."""
    addr = int(s.cuda_stream)
    if addr not in _streams:
        raise PdecError("destroy_stream: not a stream of make_stream")
    check(load().pdec_stream_destroy(_vp(addr)))       # (forgotten only once the library has released it)
    _streams.pop(addr, None)
