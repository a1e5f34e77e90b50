"""PDEagent mirror: CustomDDPGPolicy, trajectory glue, create_agent, ZeroPolicy
(src/PDEagent.jl:58-424), batched over B environments and running on libpdeconv.

Columns: one env step yields B*A (state, action) columns (A = actuators sharing the policy;
mono/global agent: one column per env).  The reference pushes A columns per step into a
CircularArraySARTTrajectory and samples `batch_size` of them per update
(src/PDEagent.jl:254-340); the batched run pushes B*A columns per step in [b][a] order, so
the next-state of column i sits at i + B*A (the reference's `inds .+ number_actuators`)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .env import _on_stream
from .nna import create_NNA

class TargetNetworkWarning(UserWarning):
    """create_agent was asked for a target-network regime other than the one that reproduces the reference's saved runs of
    the setup's experiment family (setup.reproduces_reference_with)"""


PRE_EXPERIMENT_STAGE, PRE_EPISODE_STAGE, PRE_ACT_STAGE, POST_ACT_STAGE, POST_EPISODE_STAGE, POST_EXPERIMENT_STAGE = range(6)


class ZeroPolicy:
    """src/PDEagent.jl:420-424"""

    def __init__(self, action_shape):
        self.action_shape = action_shape

    def __call__(self, env):
        return torch.zeros(env._ashape, dtype=env.dtype, device=env.device)


class RandomPolicy:
    """RLCore.RandomPolicy(action_space; rng): uniform in [-1, 1]"""

    def __init__(self, action_shape, rng=None):
        self.action_shape, self.rng = action_shape, rng or np.random.default_rng()

    def __call__(self, env):
        return torch.as_tensor(self.rng.uniform(-1, 1, env._ashape), dtype=env.dtype, device=env.device)


class NegatePolicy:
    """scripts/Fluid/setup/FluidSetup.jl:277-299: the hand-written baseline controller of the fluid script -- after the start
    steps, action[i] = clamp(-env.state[i], -1, 1) for the first length(action) entries of the state in Julia's column-major
    order (state [ns, A], action [na, A]; the device arrays [B, A, ns] / [B, A, na] flatten in the same order per
    trajectory).  No networks, nothing to train."""

    def __init__(self, action_shape, start_steps=0, start_policy=None):
        self.action_shape, self.start_steps = action_shape, start_steps
        self.start_policy = start_policy or ZeroPolicy(action_shape)
        self.update_step = 0
        self.reset_stage = POST_EPISODE_STAGE

    def __call__(self, env, learning=True, test=False):
        self.update_step += 1
        if self.update_step <= self.start_steps:
            return self.start_policy(env)
        B = env.state.shape[0]
        n = int(np.prod(env._ashape[1:]))
        return (-env.state.reshape(B, -1)[:, :n]).clamp(-1.0, 1.0).reshape(env._ashape).contiguous()


class NegateAgent:
    """create_agent_negate (FluidSetup.jl:301-322): the stage calls of RL.jl's run loop with a policy that is not trained --
    POST_EPISODE resets the step counter (:293-299), everything else is a no-op"""

    def __init__(self, policy):
        self.policy, self.trajectory = policy, None

    def __call__(self, *args):
        if len(args) == 1:
            return self.policy(args[0])
        if args[0] == POST_EPISODE_STAGE:
            self.policy.update_step = 0


def create_agent_negate(*, setup, start_steps=0, start_policy=None):
    return NegateAgent(NegatePolicy(setup.action_shape, start_steps, start_policy))


class CircularArraySARTTrajectory:
    """Device-resident replay with the reference's trace layout (state/action one `stride`
    longer than reward/terminal; Float32, src/PDEagent.jl:112-117).  Unlike RLCore's buffer the
    four traces stay aligned after wrap-around (reference quirks; the reference's misaligned buffer is emulated
    for learning-curve studies by a subclass in tests/util.py, outside the product)."""

    def __init__(self, capacity, ns, na, stride, device, reward_per_column=True):
        """On a CUDA device the stage operations are library kernels once `bind(model)` has named the handle whose
        stream they run on (row F1: pdec_replay_push_sa / _push_rt / _sample, one launch per stage); on the CPU
        (host-logic tests) they are the equivalent torch slice copies."""
        self._h = None
        self.stride = int(stride)
        self.capacity = max(self.stride, int(capacity) // self.stride * self.stride)
        kw = dict(dtype=torch.float32, device=device)
        self.state = torch.zeros((self.capacity + self.stride, ns), **kw)
        self.action = torch.zeros((self.capacity + self.stride, na), **kw)
        self.reward = torch.zeros(self.capacity, **kw)
        self.terminal = torch.zeros(self.capacity, **kw)
        self.n_sa = 0        # logical entries in the state/action traces (monotone, minus pops)
        self.n_rt = 0        # logical entries in the reward/terminal traces
        self.device = device

    def __len__(self):       # length(t) = length(t[:terminal])
        return min(self.n_rt, self.capacity)

    def bind(self, model):
        """run the stage kernels on `model`'s stream (a HipMLP of the agent)"""
        self._h, self._lib, self.stream = model.handle, model.lib, model.stream
        return self

    def _slots(self, start, n, cap):
        return (torch.arange(start, start + n, device=self.device) % cap)

    def _put(self, dst, start, src, cap):
        """rows start .. start+n-1 (mod cap) of a circular trace; one contiguous copy unless the block wraps"""
        n = src.shape[0]
        lo = start % cap
        if lo + n <= cap:
            dst[lo:lo + n].copy_(src)
        else:
            dst.index_copy_(0, self._slots(start, n, cap), src.to(dst.dtype))

    def push_sa(self, s, a):
        """PRE_ACT push of one (s, a) column per actuator (src/PDEagent.jl:254-274); a = None: the zero-action dummy of
        POST_EPISODE (:291-314)"""
        n = s.shape[0]
        cap = self.capacity + self.stride
        if self._h is not None and s.is_cuda and s.is_contiguous() and (a is None or (a.is_contiguous() and a.dtype == s.dtype)):
            _lib.check(self._lib.pdec_replay_push_sa(self._h, _lib.ptr(self.state), _lib.ptr(self.action), cap,
                                                     self.state.shape[1], self.action.shape[1], self.n_sa % cap, _lib.ptr(s),
                                                     _lib.ptr(a), n, _lib.dtype_code(s.dtype)))
        else:
            if a is None:
                a = torch.zeros((n, self.action.shape[1]), device=s.device)
            self._put(self.state, self.n_sa, s, cap)
            self._put(self.action, self.n_sa, a, cap)
        self.n_sa += n

    def pop_sa(self, n):
        self.n_sa -= n

    def push_rt(self, r, t):
        n = r.shape[0]
        self._put(self.reward, self.n_rt, r, self.capacity)
        self._put(self.terminal, self.n_rt, t, self.capacity)
        self.n_rt += n

    def push_rt_flags(self, r, done_flags, cols_per_traj, timeout):
        """POST_ACT push (src/PDEagent.jl:276-289) straight from the env step's outputs: r [n] and the per-trajectory
        int32 blow-up flags (terminal of a column = its trajectory's flag, or 1 everywhere on a time-out)"""
        n = r.shape[0]
        if self._h is not None and r.is_cuda and r.is_contiguous():
            _lib.check(self._lib.pdec_replay_push_rt(self._h, _lib.ptr(self.reward), _lib.ptr(self.terminal), self.capacity,
                                                     self.n_rt % self.capacity, _lib.ptr(r), _lib.ptr(done_flags),
                                                     int(cols_per_traj), int(bool(timeout)), n, _lib.dtype_code(r.dtype)))
            self.n_rt += n
        else:
            t = torch.ones(n) if timeout else (done_flags != 0).to(torch.float32).repeat_interleave(cols_per_traj)
            self.push_rt(r, t.to(r.device))

    def sample_device(self, seed, offset, batch_size, reuse=False):
        """pde_sample + pde_fetch! (src/PDEagent.jl:317-340) in one launch: indices from the Philox stream (seed, offset).
        reuse: write into the batch buffers of the previous call of this size (the training pipeline consumes a batch before
        it draws the next one; fixed pointers let it replay its recorded step)"""
        kw = dict(dtype=torch.float32, device=self.device)
        ns, na = self.state.shape[1], self.action.shape[1]
        out = getattr(self, "_sample_out", {}).get(batch_size) if reuse else None
        if out is None:
            out = dict(state=torch.empty((batch_size, ns), **kw), action=torch.empty((batch_size, na), **kw),
                       reward=torch.empty(batch_size, **kw), terminal=torch.empty(batch_size, **kw),
                       next_state=torch.empty((batch_size, ns), **kw))
            if reuse:
                if not hasattr(self, "_sample_out"):
                    self._sample_out = {}
                self._sample_out[batch_size] = out
        _lib.check(self._lib.pdec_replay_sample(
            self._h, _lib.ptr(self.state), _lib.ptr(self.action), _lib.ptr(self.reward), _lib.ptr(self.terminal), ns, na,
            self.capacity, self.stride, len(self), self.n_rt, int(seed), int(offset), int(batch_size), _lib.ptr(out["state"]),
            _lib.ptr(out["action"]), _lib.ptr(out["reward"]), _lib.ptr(out["terminal"]), _lib.ptr(out["next_state"]), None))
        return out

    def sample_slots(self, rng, batch_size):
        """pde_sample (src/PDEagent.jl:317-321): inds in 1:length(t)-stride -> slots (i_s, i_rt, i_sn) of the (s, a),
        (r, t) and s' entries (host numpy int64)"""
        L = len(self)
        hi = L - self.stride
        inds = rng.integers(0, hi, batch_size)
        base = max(0, self.n_rt - self.capacity)          # logical index of the oldest entry
        lg = base + inds
        return (lg % (self.capacity + self.stride), lg % self.capacity,
                (lg + self.stride) % (self.capacity + self.stride))

    def sample_slots_many(self, rng, batch_size, loops):
        """`loops` independent pde_sample draws in one vectorised call -> int array [3, loops, batch_size]
        (rows: slots of (s, a), of (r, t) and of s')"""
        hi = len(self) - self.stride
        inds = rng.integers(0, hi, (loops, batch_size))
        lg = max(0, self.n_rt - self.capacity) + inds
        cap1 = self.capacity + self.stride
        return np.stack([lg % cap1, lg % self.capacity, (lg + self.stride) % cap1])

    def sample(self, rng, batch_size):
        """pde_sample / pde_fetch! (src/PDEagent.jl:317-340)"""
        i_s, i_rt, i_sn = (torch.as_tensor(v, device=self.device) for v in self.sample_slots(rng, batch_size))
        return dict(state=self.state[i_s], action=self.action[i_s], reward=self.reward[i_rt],
                    terminal=self.terminal[i_rt], next_state=self.state[i_sn])


class CustomDDPGPolicy:
    """src/PDEagent.jl:121-209, 342-418"""

    def __init__(self, *, behavior_actor, behavior_critic, target_actor, target_critic, rng, gamma=0.99, rho=0.995,
                 batch_size=3, start_steps=6, start_policy=None, update_after=10, update_freq=1, update_loops=1,
                 reset_stage=POST_EPISODE_STAGE, act_limit=1.0, act_noise=1.2, memory_size=0, number_actuators=1,
                 quirk_target_broadcast=True, quirk_frozen_targets=True, noise_seed=0, reducer=None):
        self.behavior_actor, self.behavior_critic = behavior_actor, behavior_critic
        self.target_actor, self.target_critic = target_actor, target_critic
        self.rng = rng
        self.y, self.p = gamma, rho                       # the reference's field names (gamma, polyak rho)
        self.batch_size, self.start_steps, self.start_policy = batch_size, start_steps, start_policy
        self.update_after, self.update_freq, self.update_loops = update_after, update_freq, update_loops
        self.reset_stage, self.act_limit, self.act_noise, self.memory_size = reset_stage, act_limit, act_noise, memory_size
        self.number_actuators = number_actuators
        self.quirk = quirk_target_broadcast               # (1xBu) .+ (Bu) broadcast of r
        # The reference's Polyak loop (src/PDEagent.jl:415-417) runs over zip(Flux.params([At, Ct]), Flux.params([A, C])), and
        # both are EMPTY: src/custom_nna.jl:20 defines a `functor` of its own (not Functors.functor), so Flux sees a
        # CustomNeuralNetworkApproximator inside a Vector as a leaf without parameters.  The loop body never runs and the target
        # networks stay at their initial values for the whole training -- held by the reference's own artifact: the target actor
        # and target critic in scripts/KS/KS22/saves/agent.jld2 have exactly-zero biases and glorot-range weights after 130 340
        # updates (tests/test_replay_golden.py), and only with frozen targets does this path reproduce the reference's learning
        # curve (tests/test_gpu_training.py).  True (default) = the reference as it runs: the kernels get rho = 1 (dest = 1 * dest
        # + 0 * src); False = the loop as written (rho = self.p).
        self.quirk_frozen_targets = bool(quirk_frozen_targets)
        self.update_step = 0
        self.actor_loss = self.critic_loss = 0.0
        self.reducer = reducer                            # data-parallel gradient all-reduce (or None)
        self.use_small_update = True                      # minibatch updates of <= 16 transitions: one launch for all loops
        self._noise_seed, self._noise_off = int(noise_seed), 0
        # minibatch indices: "device" = drawn inside the update kernels from the Philox stream (sample_seed, offset),
        # "host" = drawn from self.rng as pde_sample does and handed over as index arrays
        self.sampling = "device"
        self._sample_seed, self._sample_off = int(noise_seed) ^ 0x5DEECE66D, 0
        m = behavior_actor.model
        self.lib, self.device = m.lib, m.device
        self._losses = torch.zeros(2, dtype=m.dtype, device=m.device)
        self._acting = {}                                 # dtype -> promoted acting copy of the actor
        self._noise = None
        self._actions = None
        self._set_noise_rows(m)

    def _set_noise_rows(self, model):
        """action memory: exploration noise on the driving rows only (src/PDEagent.jl:201: actions[1:end-memory_size, :] += ...);
        a property of the network object, so every caller of the acting kernels -- this policy, TrainPipeline, rollouts --
        gets it"""
        if self.memory_size:
            _lib.check(self.lib.pdec_mlp_set_noise_rows(model.handle, model.dims[-1] - self.memory_size))

    @property
    def rho_effective(self):
        """the Polyak factor the update kernels receive (see quirk_frozen_targets)"""
        return 1.0 if self.quirk_frozen_targets else float(self.p)

    # ---- acting (src/PDEagent.jl:175-209)
    def _actor_for(self, dtype, cols):
        m = self.behavior_actor.model
        if dtype == m.dtype and cols <= m.max_cols:
            return m
        key = (dtype, cols)
        if key not in self._acting:
            self._acting[key] = m.clone(dtype=dtype, max_cols=max(cols, 1))
            self._set_noise_rows(self._acting[key])
        else:
            _lib.check(self.lib.pdec_mlp_copy(self._acting[key].handle, m.handle))
        return self._acting[key]

    def __call__(self, env, learning=True, test=False):
        if learning:
            self.update_step += 1
        if self.update_step <= self.start_steps:
            return self.start_policy(env)
        s = env.state                                     # [B, A, ns] == columns [B*A, ns]
        cols = s.shape[0] * s.shape[1]
        na = self.behavior_actor.model.dims[-1]
        if self._actions is None or self._actions.shape != (cols, na) or self._actions.dtype != env.dtype:
            # two buffers, alternated: the env may adopt the returned tensor without copying (PDEenv.__call__(adopt=True))
            self._action_ring = [torch.empty((cols, na), dtype=env.dtype, device=env.device) for _ in range(2)]
            self._actions = self._action_ring[0]
        self._action_ring.reverse()
        self._actions = self._action_ring[0]
        self.act_into(s, cols, env.dtype, self._actions, learning)
        return self._actions.view(env._ashape)

    def act_into(self, state, cols, dtype, out, learning=True):
        """actor forward + randn(rng, ...) .* act_noise + clamp (:189-204) of `cols` state columns of type `dtype` into `out`
        [cols, na]: one launch, noise drawn in-kernel from the counter stream (seed, offset) -- the same numbers pdec_randn
        would produce.  An environment that computes in another type than the networks (the reference: fp64 fields, Float32
        nets) is served without a promoted copy of the actor where the library's few-column acting kernel covers the case
        (pdec_policy_act_rng_as), through a promoted clone otherwise; bit-identical either way."""
        m = self.behavior_actor.model
        na = m.dims[-1]
        served = False
        if dtype != m.dtype:
            flag = C.c_int(0)
            _lib.check(self.lib.pdec_policy_act_rng_as(m.handle, _lib.dtype_code(dtype), _lib.ptr(state), cols, float(self.act_noise),
                                                       float(self.act_limit), int(bool(learning)), self._noise_seed, self._noise_off,
                                                       _lib.ptr(out), C.byref(flag)))
            served = bool(flag.value)
        if not served:
            actor = self._actor_for(dtype, cols)
            _lib.check(self.lib.pdec_policy_act_rng(actor.handle, _lib.ptr(state), cols, float(self.act_noise),
                                                    float(self.act_limit), int(bool(learning)), self._noise_seed,
                                                    self._noise_off, _lib.ptr(out)))
        if learning:
            self._noise_off += (cols * na + 3) // 4

    # ---- update!(policy, batch) (src/PDEagent.jl:363-418), fused on the device
    def update(self, batch, before_actor_half=None):
        """batch: dict(state [Bu,ns], action [Bu,na], reward [Bu], terminal [Bu], next_state [Bu,ns]).
        before_actor_half: optional callable run between the critic half and the actor half of the update
        (e.g. a stream wait on the event of a concurrent acting kernel that still reads the actor)."""
        A, Cn, At, Ct = (self.behavior_actor.model, self.behavior_critic.model, self.target_actor.model,
                         self.target_critic.model)
        dt = Cn.dtype
        s, a, r, t, sn = (batch[k].to(dt).contiguous() for k in ("state", "action", "reward", "terminal", "next_state"))
        Bu = s.shape[0]
        L = self._losses
        oc, oa = self.behavior_critic.optimizer, self.behavior_actor.optimizer
        self._batch_keepalive = (s, a, r, t, sn)
        if self.reducer is None or not self.reducer.active:
            # single device: nothing to all-reduce, so gradient reduction, ADAM and Polyak fuse (4 launches)
            if before_actor_half is None:
                _lib.check(self.lib.pdec_ddpg_update_async(
                    A.handle, Cn.handle, At.handle, Ct.handle, _lib.ptr(s), _lib.ptr(a), _lib.ptr(r), _lib.ptr(t),
                    _lib.ptr(sn), Bu, float(self.y), self.rho_effective, int(self.quirk), float(oa.eta), float(oc.eta),
                    C.c_void_p(L.data_ptr())))
                return
            _lib.check(self.lib.pdec_ddpg_update_critic_async(
                A.handle, Cn.handle, At.handle, Ct.handle, _lib.ptr(s), _lib.ptr(a), _lib.ptr(r), _lib.ptr(t),
                _lib.ptr(sn), Bu, float(self.y), self.rho_effective, int(self.quirk), float(oc.eta), C.c_void_p(L.data_ptr())))
            before_actor_half()
            _lib.check(self.lib.pdec_ddpg_update_actor_async(
                A.handle, Cn.handle, At.handle, Ct.handle, _lib.ptr(s), Bu, self.rho_effective, float(oa.eta),
                C.c_void_p(L.data_ptr())))
            return
        self.update_critic_half(batch)
        if before_actor_half is not None:
            before_actor_half()
        self.actor_grads(batch)
        self.reducer.all_reduce(A)
        self.apply_actor()

    # ---- the pieces of the data-parallel update (reducer active), callable one by one: pipeline.TrainPipeline issues them in
    # another ORDER (the actor's ADAM step deferred behind the next critic half, its all-reduce on a side stream)
    def _batch_arrays(self, batch):
        dt = self.behavior_critic.model.dtype
        arrs = tuple(batch[k].to(dt).contiguous() for k in ("state", "action", "reward", "terminal", "next_state"))
        self._batch_keepalive = arrs
        return arrs

    def update_critic_half(self, batch):
        """critic pass + ADAM(C) + Polyak(Ct) (src/PDEagent.jl:385-400, :415-417 for the critic pair): the local fused form when
        only the policy gradient is exchanged, gradient pass -> all-reduce -> apply launch otherwise"""
        A, Cn, At, Ct = (self.behavior_actor.model, self.behavior_critic.model, self.target_actor.model,
                         self.target_critic.model)
        s, a, r, t, sn = self._batch_arrays(batch)
        Bu, L, oc = s.shape[0], self._losses, self.behavior_critic.optimizer
        if not self.reducer.reduce_critic:
            # policy-gradient-only exchange: the critic half is the local fused form (no all-reduce)
            _lib.check(self.lib.pdec_ddpg_update_critic_async(
                A.handle, Cn.handle, At.handle, Ct.handle, _lib.ptr(s), _lib.ptr(a), _lib.ptr(r), _lib.ptr(t),
                _lib.ptr(sn), Bu, float(self.y), self.rho_effective, int(self.quirk), float(oc.eta), C.c_void_p(L.data_ptr())))
            return
        _lib.check(self.lib.pdec_ddpg_critic_grads(A.handle, Cn.handle, At.handle, Ct.handle, _lib.ptr(s), _lib.ptr(a),
                                                   _lib.ptr(r), _lib.ptr(t), _lib.ptr(sn), Bu, float(self.y),
                                                   int(self.quirk), 1.0 / self.reducer.world_size, C.c_void_p(L.data_ptr())))
        self.reducer.all_reduce(Cn)
        # update!(critic) :400 fused with the critic's Polyak step :415-417 (independent of the actor)
        _lib.check(self.lib.pdec_adam_polyak_step(Cn.handle, Ct.handle, float(oc.eta), oc.beta[0], oc.beta[1],
                                                  oc.epsilon, self.rho_effective))

    def actor_grads(self, batch):
        """actor pass with the current critic + slab reduction: leaves this rank's share of the policy gradient (scaled by
        1 / world_size) in the actor's flat gradient buffer (src/PDEagent.jl:402-409)"""
        A, Cn = self.behavior_actor.model, self.behavior_critic.model
        s = self._batch_arrays(batch)[0]
        L = self._losses
        _lib.check(self.lib.pdec_ddpg_actor_grads(A.handle, Cn.handle, _lib.ptr(s), s.shape[0], 1.0 / self.reducer.world_size,
                                                  C.c_void_p(L.data_ptr() + L.element_size())))

    def apply_actor(self):
        """ADAM(A) + Polyak(At) from the (all-reduced) flat gradient buffer (:412, :415-417)"""
        A, At, oa = self.behavior_actor.model, self.target_actor.model, self.behavior_actor.optimizer
        _lib.check(self.lib.pdec_adam_polyak_step(A.handle, At.handle, float(oa.eta), oa.beta[0], oa.beta[1],
                                                  oa.epsilon, self.rho_effective))

    def small_update_ok(self):
        A, Cn = self.behavior_actor.model, self.behavior_critic.model
        return (self.use_small_update and (self.reducer is None or not self.reducer.active) and self.batch_size <= 16
                and A.dtype == torch.float32 and Cn.dtype == torch.float32 and len(A.acts) <= 4 and len(Cn.acts) <= 4)

    def update_small(self, tr, slots):
        """update_loops x update!(policy, batch) (src/PDEagent.jl:357-418) in one launch, sampling straight from the
        device-resident trajectory `tr`; slots = int array [3, loops, Bu] (rows: s/a, r/t, s' slots)"""
        A, Cn, At, Ct = (self.behavior_actor.model, self.behavior_critic.model, self.target_actor.model,
                         self.target_critic.model)
        d = torch.as_tensor(np.ascontiguousarray(slots, dtype=np.int32), device=self.device)
        oc, oa = self.behavior_critic.optimizer, self.behavior_actor.optimizer
        L = self._losses
        _lib.check(self.lib.pdec_ddpg_update_small(
            A.handle, Cn.handle, At.handle, Ct.handle, _lib.ptr(tr.state), _lib.ptr(tr.action), _lib.ptr(tr.reward),
            _lib.ptr(tr.terminal), C.c_void_p(d[0].data_ptr()), C.c_void_p(d[1].data_ptr()), C.c_void_p(d[2].data_ptr()),
            int(slots.shape[1]), int(slots.shape[2]), float(self.y), self.rho_effective, int(self.quirk), float(oa.eta),
            float(oc.eta), C.c_void_p(L.data_ptr())))
        self._slots_keepalive = d

    def update_small_rng(self, tr):
        """the same with pde_sample inside the kernel (pdec_ddpg_update_small_rng): the host passes seed + offset"""
        A, Cn, At, Ct = (self.behavior_actor.model, self.behavior_critic.model, self.target_actor.model,
                         self.target_critic.model)
        oc, oa = self.behavior_critic.optimizer, self.behavior_actor.optimizer
        _lib.check(self.lib.pdec_ddpg_update_small_rng(
            A.handle, Cn.handle, At.handle, Ct.handle, _lib.ptr(tr.state), _lib.ptr(tr.action), _lib.ptr(tr.reward),
            _lib.ptr(tr.terminal), int(self.update_loops), int(self.batch_size), self._sample_seed, self._sample_off,
            len(tr), tr.n_rt, tr.capacity, tr.stride, float(self.y), self.rho_effective, int(self.quirk), float(oa.eta),
            float(oc.eta), C.c_void_p(self._losses.data_ptr())))
        self._sample_off += (self.update_loops * self.batch_size + 3) // 4

    def losses(self):
        """(actor_loss, critic_loss) of the last update (synchronises)"""
        v = self._losses.cpu().numpy()
        self.critic_loss, self.actor_loss = float(v[0]), float(v[1])
        return self.actor_loss, self.critic_loss


class Agent:
    """RLCore.Agent(policy, trajectory) with the stage methods of src/PDEagent.jl:211-361"""

    def __init__(self, policy, trajectory):
        self.policy, self.trajectory = policy, trajectory
        self.after_push = None                # optional callable run between the PRE_ACT push of (s, a) and the update (run.py:
                                              # the overlapped run loop releases the env step there)

    def __call__(self, *args):
        if len(args) == 1:                    # agent(env) -> action
            with _on_stream(getattr(self.trajectory, "stream", None)):     # (start policies create their actions with torch)
                return self.policy(args[0])
        stage, env = args[0], args[1]
        # the torch ops of the stages run on the networks' stream, like the library kernels they are ordered with
        with _on_stream(getattr(self.trajectory, "stream", None)):
            self._stage(stage, env, args)

    def _stage(self, stage, env, args):
        p, tr = self.policy, self.trajectory
        if stage == PRE_EPISODE_STAGE:        # :237-252 pop the dummy (s, a) of the previous episode
            if len(tr) > 0 and tr.n_sa > tr.n_rt:
                tr.pop_sa(tr.stride)
        elif stage == PRE_ACT_STAGE:          # :254-274 push, then :342-361 update
            action = args[2]
            s = env.state.reshape(-1, env.state.shape[-1])
            tr.push_sa(s, action.reshape(s.shape[0], -1))
            if self.after_push is not None:
                self.after_push()
            self._maybe_update()
        elif stage == POST_ACT_STAGE:         # :276-289
            r = env.reward.reshape(-1)
            cols_per_env = r.shape[0] // env.B
            flags = getattr(env, "_done_flags", None)
            if flags is not None:             # the env step's own outputs go straight into the traces (one launch)
                tr.push_rt_flags(r, flags, cols_per_env, env.time >= env.te)
            else:
                tr.push_rt(r, env.done.to(torch.float32).repeat_interleave(cols_per_env))
        elif stage == POST_EPISODE_STAGE:     # :215-224, :291-314
            if stage == p.reset_stage:
                p.update_step = 0
            s = env.state.reshape(-1, env.state.shape[-1])
            tr.push_sa(s, None)
        elif stage == POST_EXPERIMENT_STAGE:
            if stage == p.reset_stage:
                p.update_step = 0

    def _maybe_update(self):
        p, tr = self.policy, self.trajectory
        if not (len(tr) > p.update_after * tr.stride):        # :354
            return
        if p.update_step % p.update_freq != 0:                # :355
            return
        on_device = p.sampling == "device" and getattr(tr, "_h", None) is not None
        if p.small_update_ok():
            # all update_loops minibatch updates in ONE launch; the slots of every loop (pde_sample,
            # src/PDEagent.jl:317-321) are drawn inside the kernel from the Philox stream, or here from the host rng
            if on_device:
                p.update_small_rng(tr)
            else:
                p.update_small(tr, tr.sample_slots_many(p.rng, p.batch_size, p.update_loops))   # slots: [3, loops, Bu]
            return
        for _ in range(p.update_loops):                       # :357-360
            if on_device:
                p.update(tr.sample_device(p._sample_seed, p._sample_off, p.batch_size))
                p._sample_off += (p.batch_size + 3) // 4
            else:
                p.update(tr.sample(p.rng, p.batch_size))


def resolve_target_networks(setup, requested=None, stacklevel=2):
    """-> quirk_frozen_targets for an agent on `setup`: `requested` (True / False) if given, else the regime under which this path
    reproduces the reference's saved runs of the setup's experiment family (setup.reproduces_reference_with).  A request for the
    other regime is honoured with a TargetNetworkWarning that names the measured consequence (HISTORY.md 5.1)."""
    import warnings
    family = getattr(setup, "reproduces_reference_with", "frozen")
    if requested is None:
        return family == "frozen"
    if bool(requested) != (family == "frozen"):
        warnings.warn(f"{type(setup).__name__}: quirk_frozen_targets={bool(requested)} but the reference's saved runs of this experiment "
                      f"family are reproduced only with {family} target networks (setup.reproduces_reference_with; HISTORY.md 5.1: "
                      "Keller-Segel under frozen targets saturates at return -30 in 24 of 24 seeds, the fluid diverges in 4 of 6; "
                      "KS22 / KS200 under moving targets stay a factor 3 off the reference curve)", TargetNetworkWarning,
                      stacklevel=stacklevel)
    return bool(requested)


def create_agent(*, setup, B=1, rng=None, dtype=torch.float32, device="cuda:0", start_policy=None, mono=None,
                 max_update_cols=None, reducer=None, stream=None, **overrides):
    """src/PDEagent.jl:58-119 with the setup's agent constants (KSSetup.jl:38-77).

    Target networks: `quirk_frozen_targets` defaults PER SETUP to the regime under which this path reproduces the reference's
    saved runs of that experiment family (`setup.reproduces_reference_with`: "frozen" for KSSetup -- the reference as its
    committed source runs --, "moving" for the Keller-Segel and fluid setups, whose artifacts a later session of the authors
    wrote; HISTORY.md 5.1).  Passing the other value explicitly is honoured and raises a TargetNetworkWarning that names the
    measured consequence."""
    rng = rng or np.random.default_rng(0)
    frozen = resolve_target_networks(setup, overrides.get("quirk_frozen_targets"), stacklevel=3)
    g = lambda k: overrides.get(k, getattr(setup, k))
    ns, cols_per_env = setup.state_shape
    mono = setup.mono if mono is None else mono
    na = setup.action_shape[0]
    batch_size = g("batch_size")
    max_cols = max(B * cols_per_env, batch_size, max_update_cols or 0)
    mk = lambda actor, lr: create_NNA(na=na, ns=ns, is_actor=actor, init_rng=rng,
                                      nna_scale=g("nna_scale") if actor else g("nna_scale_critic"),
                                      drop_middle_layer=g("drop_middle_layer"), learning_rate=lr, dtype=dtype,
                                      device=device, max_cols=max_cols, stream=stream)
    behavior_actor = mk(True, g("learning_rate"))
    behavior_critic = mk(False, g("learning_rate_critic"))
    target_actor = mk(True, g("learning_rate"))
    target_critic = mk(False, g("learning_rate_critic"))
    behavior_actor.copyto(target_actor)            # force sync, :76-77
    behavior_critic.copyto(target_critic)
    stride = B * cols_per_env
    policy = CustomDDPGPolicy(
        behavior_actor=behavior_actor, behavior_critic=behavior_critic, target_actor=target_actor,
        target_critic=target_critic, rng=rng, gamma=g("gamma"), rho=g("rho"), batch_size=batch_size,
        start_steps=g("start_steps"), start_policy=start_policy or ZeroPolicy(setup.action_shape),
        update_after=g("update_after"), update_freq=g("update_freq"), update_loops=g("update_loops"),
        act_limit=g("act_limit"), act_noise=g("act_noise"), memory_size=setup.memory_size,
        number_actuators=cols_per_env, reducer=reducer,
        quirk_target_broadcast=overrides.get("quirk_target_broadcast", True),
        quirk_frozen_targets=bool(frozen),
        noise_seed=overrides.get("noise_seed", 0))
    trajectory = CircularArraySARTTrajectory(g("trajectory_length") * B, ns, na, stride, torch.device(device))
    trajectory.bind(behavior_critic.model)         # stage kernels on the networks' stream (row F1)
    return Agent(policy, trajectory)
