"""PDEenv -- batched, device-resident mirror of the reference's `PDEenv` (src/PDEenv.jl:26-241).

Same field names and step semantics; the build's addition is the leading batch dimension B
(the reference has B = 1).  Tensors are torch CUDA tensors whose MEMORY matches
`[B][Julia column-major array]`, i.e. shapes are the Julia shapes reversed behind B:
    y      [B, nx]  (KS)   or [B, nx, 2]  (Keller-Segel; Julia y[2, nx])  or [B, ny, nx, 2] (2-D Keller-Segel)
    state  [B, A, ns]      (Julia state[ns, A])
    action [B, A, 1]       (Julia action[1, A])
    reward [B, A]  (mono: [B, 1]),  p [B, nx],  done [B] bool
`*_julia()` helpers return numpy copies in the Julia shapes for B = 1 (what PDEhook logs).
The whole `(env)(action)` body -- delta_action, prepare_action, integrator, reward, featurize,
blow-up test (src/PDEenv.jl:195-241) -- is ONE HIP launch (pdec_env_step)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _stream_ptr(stream):
    if stream is None:
        return None
    return C.c_void_p(stream.cuda_stream)


_FAST_STREAM_SWITCH = hasattr(torch._C, "_cuda_getCurrentStream") and hasattr(torch._C, "_cuda_setStream")


class _on_stream:
    """run the torch ops of a block on the handle's stream, so that they are ordered with the library's kernels
    (torch streams are non-blocking: work on torch's current stream is NOT ordered with another stream's kernels).
    `torch.cuda.stream(s)` costs ~8 us per enter / exit pair in Python (device-index and current-stream look-ups), and the
    B = 1 run loop enters eight of them per control step: this does the same switch with the two C calls underneath it and
    skips it altogether when `s` is current already (one process per GPU: the stream's device is the current device)."""
    __slots__ = ("s", "prev", "ctx")

    def __init__(self, stream):
        self.s, self.prev, self.ctx = stream, None, None

    def __enter__(self):
        s = self.s
        if s is None:
            return
        if not _FAST_STREAM_SWITCH:
            self.ctx = torch.cuda.stream(s)
            self.ctx.__enter__()
            return
        cur = torch._C._cuda_getCurrentStream(s.device_index)         # (stream_id, device_index, device_type)
        if cur[0] == s.stream_id:
            return
        self.prev = cur
        torch._C._cuda_setStream(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)
            self.ctx = None
        elif self.prev is not None:
            p = self.prev
            self.prev = None
            torch._C._cuda_setStream(stream_id=p[0], device_index=p[1], device_type=p[2])


class PDEenv:
    def __init__(self, setup, B=1, dtype=torch.float32, device="cuda:0", y0=None, action0=None, stream=None, history=1,
                 autoreset=None, part_streams=None):
        """part_streams: streams for the parts of the batch that the 2-D Keller-Segel and fluid environments step side by
        side (`n_part_streams` tells how many the step uses), made by the caller back to back with its pipeline streams
        (`make_streams`, include/pdeconv.h: pdec_stream_create) instead of by the library; kept alive by this object.
        autoreset (default: B > 1): a trajectory whose blow-up flag is raised by a step restarts from its initial
        condition IN that step (y, state, action rows <- y0, featurize(y0), action0; pdec_env_autoreset), its terminal
        transition having been produced; the reference (B = 1) ends the whole episode instead (src/PDEenv.jl:226-240),
        which a lock-stepped batch cannot do for one trajectory -- without the reset the blown-up trajectory would feed
        inf/NaN states into the learner until the time-out."""
        self.setup = setup
        self.B = int(B)
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PdecError("PDEenv runs on the GPU only (no CPU fallback); pass device='cuda:N'")
        self.lib = _lib.init(self.device.index or 0)
        self.te, self.t0, self.dt = setup.te, setup.t0, setup.dt
        self.oversampling = setup.oversampling
        self.max_value, self.check_max_value = setup.max_value, setup.check_max_value
        cfg = setup.env_cfg(self.B, _lib.dtype_code(dtype))
        self._h = _lib.Handle()
        self.is_fluid = bool(getattr(setup, "is_fluid", False))
        pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        if self.is_fluid:     # spectra are complex: trailing (re, im) axis
            sb, so, ab, ao, BH, BW, a2s = setup.box_tables()
            _lib.check(self.lib.pdec_fluid_env_create(
                C.byref(self._h), C.byref(cfg), BH, BW, sb.ctypes.data_as(pd), so.ctypes.data_as(pi),
                ab.ctypes.data_as(pd), ao.ctypes.data_as(pi), a2s.ctypes.data_as(pi)))
        elif getattr(setup, "is_kseg2d", False):   # u,v interleaved per cell: memory [ny][nx][2]
            sx, sy, a2s = setup.tables()
            _lib.check(self.lib.pdec_kseg2d_env_create(
                C.byref(self._h), C.byref(cfg), setup.ny, len(sx), len(sy), sx.ctypes.data_as(pi),
                sy.ctypes.data_as(pi), setup.half_window, a2s.ctypes.data_as(pi)))
        else:
            G, Ga, a2s = setup.tables()
            _lib.check(self.lib.pdec_env_create(C.byref(self._h), C.byref(cfg), G.ctypes.data_as(pd),
                                                Ga.ctypes.data_as(pd), a2s.ctypes.data_as(pi)))
        self.stream = stream
        if stream is not None:
            _lib.check(self.lib.pdec_set_stream(self._h, _stream_ptr(stream)))
        self.part_streams = None
        if part_streams is not None:
            self.set_part_streams(part_streams)
        ns, A = setup.state_shape
        self._yshape = (self.B,) + tuple(reversed(setup.y_shape)) + ((2,) if self.is_fluid else ())
        self._sshape = (self.B, A, ns)
        self._ashape = (self.B,) + tuple(reversed(setup.action_shape))
        kw = dict(dtype=dtype, device=self.device)
        if y0 is None:
            y0 = np.broadcast_to(self._to_mem(setup.y0_standard()), self._yshape[1:])
        self.y0 = self._as_batch(y0, self._yshape)
        self.action0 = (torch.zeros(self._ashape, **kw) if action0 is None else self._as_batch(action0, self._ashape))
        self.y = self.y0.clone()
        self._y_next = torch.empty_like(self.y)
        # rings: the last `history` transitions (s_t, a_t, r_t, done_t, s_{t+1}) stay valid while the NEXT
        # step is in flight, so an update on another stream can read them without a copy
        self.history = max(1, int(history))
        self._state_ring = [torch.empty(self._sshape, **kw) for _ in range(self.history + 3)]
        self._reward_ring = [torch.zeros((self.B, setup.reward_len), **kw) for _ in range(self.history + 1)]
        self._flag_ring = [torch.zeros(self.B, dtype=torch.int32, device=self.device) for _ in range(self.history + 1)]
        self._si = self._ri = 0
        self.state = self._state_ring[0]
        self.action = self.action0.clone()
        self._action_prev = self.action0.clone()
        self._adopted = set()
        self._pshape = self._yshape if self.is_fluid else (self.B,) + tuple(reversed(getattr(setup, "p_shape", (setup.nx,))))
        self.p = torch.zeros(self._pshape, **kw)
        self.reward = self._reward_ring[0]
        self._done_flags = self._flag_ring[0]
        self.prev_state = None
        self._done = torch.zeros(self.B, dtype=torch.bool, device=self.device)
        self._done_stale = False
        self.steps, self.time = 0, 0.0
        self.autoreset = (self.B > 1) if autoreset is None else bool(autoreset)
        self._state0 = torch.empty(self._sshape, **kw)
        self.reset()

    # ---- helpers

    @property
    def n_part_streams(self):
        """streams besides its own that a step of this environment uses for parts of the batch (0: none)"""
        n = C.c_int(0)
        _lib.check(self.lib.pdec_env_part_streams(self._h, C.byref(n)))
        return n.value

    def set_part_streams(self, streams):
        """hands the caller's part streams to the environment (pdec_env_set_part_streams); same results bit for bit"""
        streams = list(streams)
        arr = (C.c_void_p * max(len(streams), 1))(*[s.cuda_stream for s in streams])
        _lib.check(self.lib.pdec_env_set_part_streams(self._h, arr, len(streams)))
        self.part_streams = streams

    def _to_mem(self, a):
        """Julia-shaped host array -> memory image (column-major == reversed axes); complex fields get a
        trailing (re, im) axis"""
        a = np.asarray(a)
        if a.ndim == 3 and not self.is_fluid:     # Julia y[2, nx, ny] (oracle: [2, ny, nx]) -> memory [ny][nx][2]
            return np.ascontiguousarray(np.moveaxis(np.asarray(a, dtype=np.float64), 0, -1))
        if np.iscomplexobj(a) or self.is_fluid:
            a = np.asarray(a, dtype=np.complex128)
            a = np.swapaxes(a, -1, -2) if a.ndim >= 2 else a
            return np.stack([a.real, a.imag], axis=-1)
        a = np.asarray(a, dtype=np.float64)
        return a.T if a.ndim == 2 else a

    def _as_batch(self, a, shape):
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=self.dtype)
        else:
            a = np.asarray(a)
            if np.iscomplexobj(a):        # Julia-shaped complex field(s) [.., ny, nx]
                a = self._to_mem(a)
            t = torch.as_tensor(np.array(a, copy=True), dtype=self.dtype, device=self.device)
        if t.dim() == len(shape) - 1:
            t = t.unsqueeze(0).expand(shape)
        if tuple(t.shape) != tuple(shape):
            raise _lib.PdecError(f"expected shape {shape} (or without the batch axis), got {tuple(t.shape)}")
        return t.contiguous().clone()

    @property
    def handle(self):
        return self._h

    @property
    def done(self):
        """per-trajectory episode-end flags (src/PDEenv.jl:224-240), materialised on demand so that a control step
        issues no kernel besides the fused env step"""
        if self._done_stale:
            with _on_stream(self.stream):          # ordered behind the env step that wrote the flags
                if self.time >= self.te:
                    self._done.fill_(True)
                else:
                    torch.ne(self._done_flags, 0, out=self._done)
            self._done_stale = False
        return self._done

    def set_terminal_out(self, buf):
        """have every later step also write per-column terminal flags ([B, A] of the env dtype) for the DDPG batch"""
        self._terminal_out = buf
        _lib.check(self.lib.pdec_env_set_terminal_out(self._h, _lib.ptr(buf)))

    @property
    def delta_action(self):
        return self.action - self._action_prev

    # ---- RLBase surface (src/PDEenv.jl:172-181)
    def state_space_size(self):
        return self.setup.state_shape

    def action_space_size(self):
        return self.setup.action_shape

    def is_terminated(self):
        """Episode end.  Time-out is common to the lock-stepped batch; a blown-up trajectory (done[b]) ends the episode
        only when B == 1 (the reference's case) -- with autoreset it has already restarted from its initial condition,
        without autoreset the batch goes on until every trajectory has blown up."""
        if self.time >= self.te:
            return True
        if self.B > 1 and self.autoreset:
            return False
        d = self.done
        with _on_stream(self.stream):              # the read-back waits for the env's stream, not for torch's current one
            return bool(d.all().item()) if self.B > 1 else bool(d[0].item())

    def set_y0(self, y0):
        """new initial condition (PDEhook's PRE_EPISODE random re-initialisation, src/PDEhook.jl:42-49): env.y0, env.y,
        env.state and the image a per-trajectory reset restores"""
        with _on_stream(self.stream):
            self.y0 = y0 if (isinstance(y0, torch.Tensor) and tuple(y0.shape) == self._yshape and y0.dtype == self.dtype
                             and y0.is_contiguous()) else self._as_batch(y0, self._yshape)
            self.y.copy_(self.y0)
            _lib.check(self.lib.pdec_featurize(self._h, _lib.ptr(self.y), None, _lib.ptr(self._state0)))
            self.state.copy_(self._state0)

    # ---- stand-alone closures (each one launch)
    def featurize(self, y=None, prev_state=None, action=None):
        """featurize(y0, t0) / featurize(; env) (KSSetup.jl:190-229).  `action` [B, A, 1 + memory_size]: the source of the
        action-memory rows (memory_size > 0; None = the reset form, zeros)"""
        y = self.y if y is None else y
        out = torch.empty(self._sshape, dtype=self.dtype, device=self.device)
        _lib.check(self.lib.pdec_featurize_action(self._h, _lib.ptr(y), _lib.ptr(prev_state), _lib.ptr(action), _lib.ptr(out)))
        return out

    def prepare_action(self, action=None):
        action = self.action if action is None else action
        out = torch.empty(self._pshape, dtype=self.dtype, device=self.device)
        _lib.check(self.lib.pdec_actuate(self._h, _lib.ptr(action), _lib.ptr(out)))
        return out

    def reward_function(self, y=None, action=None, action_prev=None):
        y = self.y if y is None else y
        action = self.action if action is None else action
        action_prev = self._action_prev if action_prev is None else action_prev
        out = torch.empty((self.B, self.setup.reward_len), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.pdec_reward(self._h, _lib.ptr(y), _lib.ptr(action), _lib.ptr(action_prev), _lib.ptr(out)))
        return out

    def do_step(self, y=None, p=None):
        """do_step(env): integrator only (KSSetup.jl:130-160 / KellerSegelSetup.jl:234-239)"""
        y = self.y if y is None else y
        p = self.p if p is None else p
        out = torch.empty_like(y)
        flags = torch.empty(self.B, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.pdec_pde_step(self._h, _lib.ptr(y), _lib.ptr(p), _lib.ptr(out), _lib.ptr(flags)))
        return out, flags

    def rhs(self, y, p):
        out = torch.empty_like(y)
        _lib.check(self.lib.pdec_rhs_eval(self._h, _lib.ptr(y), _lib.ptr(p), _lib.ptr(out)))
        return out

    def random_init(self, seed, offset, out=None):
        """generate_random_init() of the 1-D setups on the device (pdec_env_random_init; scripts/KS/setup/KSSetup.jl:288-298,
        scripts/Keller-Segel/setup/KellerSegelSetup.jl:373-384): fills `out` (default: a new tensor shaped like env.y) from
        the Philox stream (seed, offset) and returns the number of counters consumed"""
        out = torch.empty_like(self.y) if out is None else out
        _lib.check(self.lib.pdec_env_random_init(self._h, int(seed), int(offset), _lib.ptr(out)))
        self._last_random_init = out
        nc = 8 if self.setup.y_shape == (self.setup.nx,) else 2 * int(np.ceil(self.setup.Lx / 3))
        return self.B * ((nc + 3) // 4)

    # ---- reset!(env), src/PDEenv.jl:183-193
    def reset(self):
        with _on_stream(self.stream):
            self.y.copy_(self.y0)
            _lib.check(self.lib.pdec_featurize(self._h, _lib.ptr(self.y), None, _lib.ptr(self.state)))
            self._state0.copy_(self.state)
            self.prev_state = None
            if self.action.data_ptr() in self._adopted:      # never write into a caller-owned buffer
                self.action = torch.empty_like(self.action0)
            if self._action_prev.data_ptr() in self._adopted:
                self._action_prev = torch.empty_like(self.action0)
            self.action.copy_(self.action0)
            self._action_prev.copy_(self.action0)
            self.p = self.prepare_action(self.action0)
            self.steps, self.time = 0, 0.0
            self.reward.zero_()
            self._done.zero_()
            self._done_stale = False

    def reset_episode(self):
        """reset!(env) for a pipelined caller: like reset(), but the initial state goes into the NEXT slot of the state
        ring, so the (s, s') tensors of the transitions still in flight (e.g. an update running on another stream) stay
        intact, and nothing is allocated or synchronised"""
        self.y.copy_(self.y0)
        self._si = (self._si + 1) % len(self._state_ring)
        st = self._state_ring[self._si]
        _lib.check(self.lib.pdec_featurize(self._h, _lib.ptr(self.y), None, _lib.ptr(st)))
        self._state0.copy_(st)
        self.state, self.prev_state = st, None
        if getattr(self, "_action_reset", None) is None:
            self._action_reset = self.action0.clone()
            self._adopted.add(self._action_reset.data_ptr())      # read-only from now on
        self._action_prev = self.action = self._action_reset
        self.steps, self.time = 0, 0.0
        self._done.zero_()
        self._done_stale = False

    # ---- (env::PDEenv)(action), src/PDEenv.jl:195-241
    def __call__(self, action, adopt=False):
        """adopt=True: keep a reference to `action` instead of copying it (no D2D copy on the step's critical
        path); the caller must then leave that buffer untouched until the step AFTER the next one has been
        issued (env.action / env.delta_action read it), e.g. by alternating two buffers."""
        if action.dtype != self.dtype or not action.is_contiguous() or tuple(action.shape) != self._ashape:
            with _on_stream(self.stream):
                action = action.to(self.dtype).reshape(self._ashape).contiguous()
            adopt = False
        if adopt:
            self._action_prev, self.action = self.action, action
        else:
            self._action_prev, self.action = self.action, self._action_prev
            with _on_stream(self.stream):
                if self.action.data_ptr() == action.data_ptr() or self.action.data_ptr() in self._adopted:
                    self.action = torch.empty_like(action)     # never write into a caller-owned (adopted) buffer
                self.action.copy_(action)
        if adopt:
            self._adopted.add(action.data_ptr())
        self._si = (self._si + 1) % len(self._state_ring)
        self._ri = (self._ri + 1) % len(self._reward_ring)
        state_next = self._state_ring[self._si]
        self.reward, self._done_flags = self._reward_ring[self._ri], self._flag_ring[self._ri]
        _lib.check(self.lib.pdec_env_step(
            self._h, _lib.ptr(self.y), _lib.ptr(self.action), _lib.ptr(self._action_prev), _lib.ptr(self.state),
            _lib.ptr(self._y_next), _lib.ptr(self.p), _lib.ptr(state_next), _lib.ptr(self.reward),
            _lib.ptr(self._done_flags)))
        self.y, self._y_next = self._y_next, self.y
        self.prev_state, self.state = self.state, state_next
        self.steps += 1
        self.time += self.dt
        self._done_stale = True
        if self.autoreset:
            own_action = self.action.data_ptr() not in self._adopted
            _lib.check(self.lib.pdec_env_autoreset(
                self._h, _lib.ptr(self._done_flags), _lib.ptr(self.y), _lib.ptr(self.y0), _lib.ptr(self.state),
                _lib.ptr(self._state0), _lib.ptr(self.action) if own_action else None,
                _lib.ptr(self.action0) if own_action else None, _lib.ptr(self.reward)))

    def set_simd_sharing(self, on=True):
        """Launch the fused KS step in its 64-VGPR form (constants in LDS), whose waves can share a SIMD with the f32-MFMA
        critic pass -- for callers that run the step beside the update passes on a second stream (TrainPipeline does).
        Returns True when this environment has such a form (KS CNAB2, N = 256, fp32)."""
        import ctypes as C
        eff = C.c_int()
        _lib.check(self.lib.pdec_env_set_simd_sharing(self.handle, 1 if on else 0, C.byref(eff)))
        return bool(eff.value)

    # ---- T control steps without returning to the host
    def rollout(self, actor, T, act_noise=0.0, act_limit=1.0, learning=False, seed=0, offset=0, log=False):
        """for t in 1:T; action = policy(env); env(action); end  (src/PDEagent.jl:175-209 + src/PDEenv.jl:195-241)
        enqueued in ONE library call: actor forward (+ exploration noise when `learning`), clamp, fused env step,
        device-side reward accumulation and -- with log=True -- the per-step rows PDEhook records
        (src/PDEhook.jl:54-62).  `actor` is a HipMLP of the environment's dtype on the environment's stream.
        Returns device tensors: reward_sum [B, A], done_step [B] (first step that raised `done`, -1 = none) and the logs
        y [T, ...], p, action, reward.  env.y / state / action / steps / time advance by T steps."""
        T = int(T)
        kw = dict(dtype=self.dtype, device=self.device)
        with _on_stream(self.stream):
            out = dict(reward_sum=torch.zeros((self.B, self.setup.reward_len), **kw),
                       done_any=torch.zeros(self.B, dtype=torch.int32, device=self.device),
                       done_step=torch.zeros(self.B, dtype=torch.int32, device=self.device))
            if log:
                out.update(y=torch.empty((T,) + self._yshape, **kw), p=torch.empty((T,) + self._pshape, **kw),
                           action=torch.empty((T,) + self._ashape, **kw),
                           reward=torch.empty((T, self.B, self.setup.reward_len), **kw))
            if self.action.data_ptr() in self._adopted:      # never write into a caller-owned buffer
                self.action = self.action.clone()
        state = self.state
        # the rollout runs on the environment's stream; an actor that lives on another one (an agent created with its own
        # stream for the overlapped run loop) is moved over for the call, ordered behind whatever last wrote its parameters
        astream = getattr(actor, "stream", None)
        moved = (astream is not None and self.stream is not None and astream.cuda_stream != self.stream.cuda_stream)
        if moved:
            self.stream.wait_stream(astream)
            _lib.check(self.lib.pdec_set_stream(actor.handle, _stream_ptr(self.stream)))
        try:
            _lib.check(self.lib.pdec_rollout(
                self._h, actor.handle, T, _lib.ptr(self.y), _lib.ptr(state), _lib.ptr(self.action), float(act_noise),
                float(act_limit), int(bool(learning)), int(seed), int(offset), _lib.ptr(out["reward_sum"]),
                _lib.ptr(out.get("y")), _lib.ptr(out.get("p")), _lib.ptr(out.get("action")), _lib.ptr(out.get("reward")),
                _lib.ptr(out["done_any"]), _lib.ptr(out["done_step"])))
        finally:
            if moved:
                _lib.check(self.lib.pdec_set_stream(actor.handle, _stream_ptr(astream)))
                astream.wait_stream(self.stream)
        self.prev_state = None
        self.steps += T
        for _ in range(T):                 # the same floating-point sum as T single steps (50 x 0.1 != 5.0)
            self.time += self.dt
        with _on_stream(self.stream):
            if self.time >= self.te:       # done = time >= te or blow-up (src/PDEenv.jl:227), as after a step
                self._done.fill_(True)
            else:
                torch.ne(out["done_any"], 0, out=self._done)
        self._done_stale = False
        return out

    # ---- Julia-shaped host views for B == 1 (what PDEhook logs, src/PDEhook.jl:54-62)
    def y_julia(self, b=0):
        a = self.y[b].detach().cpu().numpy().astype(np.float64)
        if self.is_fluid:
            return (a[..., 0] + 1j * a[..., 1]).T
        if a.ndim == 3:
            return np.moveaxis(a, -1, 0)
        return a.T if a.ndim == 2 else a

    def state_julia(self, b=0):
        return self.state[b].detach().cpu().numpy().astype(np.float64).T

    def action_julia(self, b=0):
        a = self.action[b].detach().cpu().numpy().astype(np.float64)
        return a.T if a.ndim == 2 else a

    def close(self):
        if self._h is not None:
            self.lib.pdec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
