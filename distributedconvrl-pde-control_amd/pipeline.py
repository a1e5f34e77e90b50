"""The batched training step as a two-stream pipeline, eager or replayed from captured HIP graphs.

One control step keeps the reference's stage order -- act -> update -> env step (RL.jl's run loop through
src/PDEagent.jl:175-209, :342-418 and src/PDEenv.jl:195-241):

    env stream:     wait update_{k-1}  ->  act_k (actor forward + exploration noise + clamp)  ->  env_k (fused env step)
    update stream:  wait act_{k-1}     ->  update_k (critic half, actor half: 4 launches) on the transition of step k - LAG

The update needs nothing of step k (DDPG is off-policy; the reference samples its minibatches from a 150k-deep replay,
src/PDEagent.jl:317-340) and act_k reads a published copy of the actor that update_k does not write, so update_k runs
back to back behind update_{k-1} beside act_k / env_k; act_{k+1} follows update_k, exactly as `agent(env)` follows
`update!` in the run loop.  (update_k must follow act_{k-1}, the last reader of the image its actor half republishes;
env_{k-LAG}, its batch, precedes act_{k-1} on the env stream.)

Data-parallel runs (N > 1, policy-gradient-only exchange, frozen target networks): the all-reduce of the actor's gradient and
the ADAM launch that consumes it leave the update stream (`stream_ar`):

    update stream:  critic half_k -> [wait apply_{k-1}] actor pass_k -> slab reduction_k ---------> critic half_{k+1} -> ...
    side stream:                                                        wait reduction_k -> all-reduce_k -> ADAM(actor)_k

The critic half reads the TARGET actor only (a' = At(s'), src/PDEagent.jl:385), which never moves (quirk_frozen_targets), so
critic half_{k+1} does not depend on ADAM(actor)_k and hides the collective; the next readers of the updated actor -- act_{k+1} on
the env stream and actor pass_{k+1} -- wait for the event on the ADAM launch.  Same kernels, same arguments, same order of
arithmetic as the in-chain order: bit-identical (tests/test_gpu_pipeline.py, tests/test_aa_multirank_gpu.py).

Every buffer a step touches is a pure function of the step counter k (rings indexed by k mod 2 / 3 / 6) and every
per-step scalar lives on the device (noise counter: pdec_policy_act_rng_dev; ADAM beta powers), so the launch arguments
of step k + 6 are those of step k: chunks of 24 / 6 / 1 consecutive steps starting at each of the six ring phases are
captured ONCE into HIP graphs (pdec_capture_begin / _end) and replayed with one host call per chunk instead of ~12 ctypes
calls per step.  Inside a chunk the cross-stream events above become graph edges; a chunk forks the update branch off
the env stream at its first step and joins it at its last.  The first and the last step of an episode
(initial-condition pointers, time-out terminal flags; te / dt + 1 = 51 steps in scripts/KS/KS22) are issued eagerly.  Eager and replayed runs enqueue the same kernels with the same arguments and are
bit-identical (tests/test_gpu_agent.py)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

PERIOD = 6            # lcm of the ring lengths (2: y / published actor / beta-power and noise-counter slots; 3; 6)


class _Event:
    """device-scope event of the library (pdec_event_create: no system-scope fence), torch.cuda.Event-like surface"""

    def __init__(self, lib):
        self.lib, self.h = lib, _lib.Handle()
        _lib.check(lib.pdec_event_create(C.byref(self.h)))

    def record(self, stream):
        _lib.check(self.lib.pdec_event_record(self.h, C.c_void_p(stream.cuda_stream)))

    def wait(self, stream):
        _lib.check(self.lib.pdec_stream_wait_event(C.c_void_p(stream.cuda_stream), self.h))

    def __del__(self):
        try:
            self.lib.pdec_destroy(self.h)
        except Exception:
            pass


class _TorchEvent:
    """the same surface on a default torch event (diagnostic: PDEC_TORCH_EVENTS=1)"""

    def __init__(self, lib=None):
        self.ev = torch.cuda.Event()

    def record(self, stream):
        self.ev.record(stream)

    def wait(self, stream):
        stream.wait_event(self.ev)


class TrainPipeline:
    def __init__(self, env, agent, lag=2, episode_steps=51, stream_env=None, stream_upd=None, use_graphs=True,
                 chunks=(24, 6, 1), use_replay=False, noise_seed=1234, kick_env_after_critic=None, stream_ar=None,
                 ar_off_chain=None):
        """env: PDEenv on `stream_env`; agent: create_agent(..., stream=stream_upd).  lag: the update of step k trains on
        the transition of step k - lag (>= 1).  episode_steps: lock-stepped episodes of that many control steps (0: one
        endless episode): the last transition is terminal (done = time >= te, src/PDEenv.jl:227) and the next step starts
        from env.y0 (reset!, :183-193).  use_replay: route the update through the device-resident replay (row F1): every
        transition is pushed (src/PDEagent.jl:254-289) and the update trains on B * A transitions drawn from it
        (:317-340) instead of the B * A fresh ones (eager only).  stream_ar: a third stream for the gradient all-reduce + the
        actor's ADAM launch of a data-parallel run (module docstring); ar_off_chain: None = whenever that order is legal
        (split update, policy-gradient-only exchange, frozen targets, stream_ar given), False = keep the collective on the update
        stream, True = insist (raises when illegal)."""
        self.env, self.agent, self.policy = env, agent, agent.policy
        self.lib = env.lib
        self.LAG = max(1, int(lag))
        assert self.LAG in (1, 2), "rings hold the last 3 transitions: lag 1 or 2"
        self.E = int(episode_steps)
        self.s_env = stream_env if stream_env is not None else env.stream
        self.s_upd = stream_upd if stream_upd is not None else self.policy.behavior_critic.model.stream
        if self.s_env is None or self.s_upd is None:
            raise _lib.PdecError("TrainPipeline needs explicit (non-default) streams for the environment and the networks")
        self.serial = self.s_env.cuda_stream == self.s_upd.cuda_stream
        self.use_replay = bool(use_replay)
        reducer = self.policy.reducer
        self.multi_rank = reducer is not None and reducer.active      # the split update sequence (N > 1, or forced)
        # the all-reduce + ADAM(actor) beside the next critic half instead of in front of it: legal when that critic half cannot
        # see the actor's update -- it reads the target actor, so the targets must be frozen -- and the critic is not exchanged
        self._ar_legal = (self.multi_rank and not reducer.reduce_critic and bool(self.policy.quirk_frozen_targets)
                          and stream_ar is not None and stream_ar.cuda_stream not in (self.s_upd.cuda_stream, self.s_env.cuda_stream))
        if ar_off_chain and not self._ar_legal:
            raise _lib.PdecError("TrainPipeline(ar_off_chain=True) needs an active reducer with reduce_critic=False, frozen target "
                                 "networks (quirk_frozen_targets) and a stream_ar that is neither the env nor the update stream")
        self._stream_ar = stream_ar
        self.ar_off_chain = self._ar_legal if ar_off_chain is None else bool(ar_off_chain)
        self.s_ar = stream_ar if self.ar_off_chain else None
        # a recorded step / a graph holds POINTERS: policy.update must hand the ring tensors through unchanged, which it
        # does only when no dtype conversion makes a temporary (`.to(dt).contiguous()` is the identity then)
        self._batch_aliases = env.dtype == self.policy.behavior_critic.model.dtype
        self.use_graphs = bool(use_graphs) and not self.use_replay and not self.multi_rank and self._batch_aliases
        self.chunks = tuple(sorted({int(c) for c in chunks if c == 1 or c % PERIOD == 0} | {1}, reverse=True))
        setup, B = env.setup, env.B
        ns, A = setup.state_shape
        self.cols = B * A
        self.ns, self.na = ns, setup.action_shape[0]
        dt, dev = env.dtype, env.device
        kw = dict(dtype=dt, device=dev)
        import os
        empty = (lambda shape, **k: torch.full(shape, float("nan"), **k)) if os.environ.get("PDEC_POISON") else torch.empty
        with torch.cuda.stream(self.s_env):
            self.ybuf = [empty(env._yshape, **kw) for _ in range(2)]
            self.sring = [empty(env._sshape, **kw) for _ in range(PERIOD)]
            self.aring = [torch.zeros(env._ashape, **kw) for _ in range(3)]
            self.rring = [torch.zeros((B, setup.reward_len), **kw) for _ in range(3)]
            self.fring = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(3)]
            self.tring = [torch.zeros((B, setup.reward_len), **kw) for _ in range(3)]      # per-column terminal flags
            self.pbuf = torch.zeros(env._pshape, **kw)
            self.rbar = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in range(3)]  # batch-mean reward of step k
            self.azero = torch.zeros(env._ashape, **kw)                                    # action0 of a fresh episode
            self.state0 = empty(env._sshape, **kw)
        # the reference's reward broadcast (quirk) needs the batch-mean reward: reduced on the env stream
        # right behind the env step instead of by every workgroup of the critic pass
        self.pre_rbar = bool(self.policy.quirk) and dt == torch.float32 and not self.use_replay
        # ... and for the fused KS step + 3-layer fused critic without any extra launch: the env step leaves one reward sum
        # per workgroup, the critic pass adds them
        self.rpart, self.n_rpart = None, 0
        if self.pre_rbar:
            npart = C.c_int()
            probe = torch.zeros(B, dtype=torch.float32, device=dev)
            ok = self.lib.pdec_env_set_reward_partials_out(env.handle, _lib.ptr(probe), C.byref(npart)) == 0
            hc = self.policy.behavior_critic.model.handle
            ok = ok and self.lib.pdec_ddpg_set_reward_partials(hc, _lib.ptr(probe), npart.value) == 0    # refused unless fused 3-layer
            self.lib.pdec_ddpg_set_reward_partials(hc, None, 0)
            self.lib.pdec_env_set_reward_partials_out(env.handle, None, None)
            if ok:
                self.n_rpart = npart.value
                self.rpart = [torch.zeros(self.n_rpart, dtype=torch.float32, device=dev) for _ in range(3)]
        import os
        Ev = _TorchEvent if os.environ.get("PDEC_TORCH_EVENTS") == "1" else _Event
        self.ev_fork = Ev(self.lib)
        self.ev_act = [Ev(self.lib), Ev(self.lib)]                  # act_k issued (env stream)
        self.ev_upd = [Ev(self.lib), Ev(self.lib)]                  # update_k issued (update stream)
        self.ev_graph = Ev(self.lib)                                # tail of the last graph launch (env stream)
        self.ev_mid = Ev(self.lib)                                  # critic half of update_k done (update stream)
        self.ev_push = [Ev(self.lib), Ev(self.lib)]                 # replay pushes of step k done (env stream)
        self.ev_samp = Ev(self.lib)                                 # replay sample of update_k done (update stream)
        self.ev_red = Ev(self.lib)                                  # actor gradient of update_k reduced (update stream) -> side stream
        self.ev_env = [Ev(self.lib), Ev(self.lib)]                  # env_k done (env stream; collective off the chain only)
        # collective off the chain, two streams, LAG >= 2: both cross-stream waits of the update stream sit between the halves
        # of an update (between_halves) instead of one there and one at the top of the step
        self._mid_waits = self.ar_off_chain and not self.serial and self.LAG >= 2 and not self.use_replay
        self._samp_pending = False
        self._after_graph = False
        self.tick = 0             # control steps issued so far: every buffer of step k is indexed by k mod 2 / 3 / 6
        self.ep_start = 0         # tick of the first step of the current episode
        self.noise_seed = int(noise_seed)
        self.actor = self.policy.behavior_actor.model
        yes = C.c_int()
        _lib.check(self.lib.pdec_mlp_acts_on_published_copy(self.actor.handle, C.byref(yes)))
        # an actor whose acting kernel reads its parameters in place (2-layer / generic nets) must not be rewritten by the
        # update's actor half while act_k still runs: that half then waits for act_k (the fused 3-layer path reads a
        # published copy instead and needs no such edge)
        self.act_in_place = not bool(yes.value)
        # start env_k behind the critic half of update_k instead of beside the critic pass (see _issue); for the
        # reference-shaped 2-layer nets the env branch is as long as the whole update, so it is not held back there
        # (device replay: the sample of update_{k+1} waits for the pushes behind env_k, so a step held back behind the critic
        # half stalls the update stream -- 158.7 vs 129.1 us per step, r03cm -- and the step is enqueued at once there)
        self.kick_env_after_critic = ((not self.act_in_place and not self.use_replay) if kick_env_after_critic is None
                                      else bool(kick_env_after_critic))
        if os.environ.get("PDEC_KICK") in ("0", "1"):               # diagnostic override
            self.kick_env_after_critic = os.environ["PDEC_KICK"] == "1"
        # events attached to the reduction launches instead of recorded behind them (see _issue); PDEC_STOP_EVENTS=0: records
        # (N > 1: the launch that applies the update after the all-reduce carries the event, pdec_adam_polyak_step)
        self.stop_events = (self.rpart is not None and Ev is _Event and not self.serial
                            and os.environ.get("PDEC_STOP_EVENTS", "1") != "0")
        # The PDE step beside the fused 3-layer passes.  Round 2: its register form (92 VGPRs) could not share a SIMD with two
        # waves of the 222-VGPR critic pass, so the pipeline asked for the 64-VGPR form (csrc/env.hip, SHARE) at priority 3.
        # Round 3: the passes are bounded to 208 allocated VGPRs (2 x 208 + 96 = 512), the register form fits beside them, and
        # at its own priority 1 it takes less from the passes: 117.7 -> 111.4 us per control step (r03bc).  PDEC_SHARE=1 asks for
        # the 64-VGPR form again.
        self.simd_sharing = False
        if not self.serial and self.rpart is not None and os.environ.get("PDEC_SHARE", "0") == "1":
            eff = C.c_int()
            _lib.check(self.lib.pdec_env_set_simd_sharing(env.handle, 1, C.byref(eff)))
            self.simd_sharing = bool(eff.value)
        self._sp_env, self._sp_upd = C.c_void_p(self.s_env.cuda_stream), C.c_void_p(self.s_upd.cuda_stream)
        self._sp_ar = C.c_void_p(self.s_ar.cuda_stream) if self.s_ar is not None else None
        self.graphs = {}          # (chunk, pos) -> graph handle
        self._progs = {}          # ring phase -> recorded library calls of an interior eager step
        # the recorded-call replay re-issues LIBRARY calls only: with torch events (PDEC_TORCH_EVENTS=1) the cross-stream
        # records / waits are torch calls it would not see, so it is off for that event type
        # N > 1: the gradient all-reduce is one more recorded call (a library call with NativeGradReducer, a noted torch call
        # with GradReducer -- _Lib.note), so the multi-rank pipeline issues its steps the same fast way
        # (device replay: the batch is the trajectory's fixed fp32 sample buffers, handed through unchanged to fp32 nets)
        aliases = (self.policy.behavior_critic.model.dtype == torch.float32) if self.use_replay else self._batch_aliases
        self.fast_eager = os.environ.get("PDEC_FAST_EAGER", "1") == "1" and Ev is _Event and aliases
        self._key = None          # per-step scalars baked into _progs / the graphs (see _scalar_key)
        self._recapture = False
        self._captured = False
        self.n_graph_launches = self.n_eager_steps = 0
        self.reset_from(env.y0)

    def set_ar_order(self, off_chain):
        """switch between the two places of the gradient all-reduce (module docstring) in a running pipeline: drains the
        streams, drops the recorded steps.  Both orders compute the same numbers; a caller times both and keeps the faster
        (bench.py: a collective of a few microseconds is cheaper left on the chain than two stream hops are)."""
        off_chain = bool(off_chain)
        if off_chain and not self._ar_legal:
            raise _lib.PdecError("set_ar_order(True): not legal for this pipeline (see TrainPipeline(ar_off_chain=...))")
        self.sync()
        self.ar_off_chain = off_chain
        self.s_ar = self._stream_ar if off_chain else None
        self._sp_ar = C.c_void_p(self.s_ar.cuda_stream) if self.s_ar is not None else None
        self._mid_waits = self.ar_off_chain and not self.serial and self.LAG >= 2 and not self.use_replay
        self._mid_wait_prev = False
        self._progs = {}

    # ------------------------------------------------------------------ state
    def reset_from(self, y0):
        """(re)start: the next step is the first of an episode from initial condition y0 ([B, ...] device tensor)"""
        env = self.env
        with torch.cuda.stream(self.s_env):
            if y0.data_ptr() != env.y0.data_ptr():
                env.y0.copy_(y0)
            _lib.check(self.lib.pdec_featurize(env.handle, _lib.ptr(env.y0), None, _lib.ptr(self.state0)))
        self.ep_start = self.tick
        if self.tick > 0:
            # a restart in the middle of an episode: the transitions of the steps before it must not be trained on -- the
            # first step of the new episode overwrites sring[tick % 6], which is the next_state of transition tick - 1 and
            # would be bootstrapped across the reset with terminal = 0 -- and the recorded interior steps are dropped
            self._first_tick = self.tick
            self._progs = {}
            tr = self.agent.trajectory
            if self.use_replay and tr.n_rt > 0 and tr.n_sa == tr.n_rt:
                # device-replay route: transition tick-1 is already in the ring with terminal = 0 and no
                # POST_EPISODE dummy follows it, so the first (s, a) row of the new episode becomes its next_state and later
                # samples would bootstrap across the reset.  Its rows are cut off instead: terminal = 1 (the layout a regular
                # episode end leaves once its dummy row has been popped and overwritten).  On the env stream, behind the pushes.
                with torch.cuda.stream(self.s_env):
                    lo = (tr.n_rt - self.cols) % tr.capacity
                    tr.terminal[lo:lo + self.cols].fill_(1.0)

    @property
    def y(self):
        return self.ybuf[self.tick % 2]

    @property
    def state(self):
        return self.sring[self.tick % PERIOD]

    # ------------------------------------------------------------------ one step, issued call by call
    def _issue(self, k, chunk_first=False, chunk_last=False):
        """enqueue control step k.  chunk_first / chunk_last: first / last step of a chunk being CAPTURED (fork / join of
        the update branch); both False for a step issued eagerly."""
        env, pol, lib, L = self.env, self.policy, self.lib, _lib
        first, last = self._first_last(k)
        y_in, y_out = self.ybuf[k % 2], self.ybuf[(k + 1) % 2]
        s_in, s_out = self.sring[k % PERIOD], self.sring[(k + 1) % PERIOD]
        act, act_prev = self.aring[k % 3], self.aring[(k - 1) % 3]
        rew, flags, term = self.rring[k % 3], self.fring[k % 3], self.tring[k % 3]
        capturing = chunk_first or chunk_last or self._capturing
        mid_wait_prev, self._mid_wait_prev = self._mid_wait_prev, False
        if not self.serial:
            if chunk_first:                       # fork: everything before this chunk is ordered before it on the env stream
                self.ev_fork.record(self.s_env)
                self.ev_fork.wait(self.s_upd)
            elif capturing:
                self.ev_act[(k - 1) % 2].wait(self.s_upd)
                self.ev_upd[(k - 1) % 2].wait(self.s_env)
            else:                                 # eager
                if self._after_graph:             # the update branch of the graph just launched ran on the env stream
                    self.ev_graph.wait(self.s_upd)
                    self._after_graph = False
                elif k > 0:
                    # (collective off the chain: update_{k-1} waited for env_{k-2} between its halves, see between_halves)
                    if not mid_wait_prev:
                        self.ev_act[(k - 1) % 2].wait(self.s_upd)
                    self.ev_upd[(k - 1) % 2].wait(self.s_env)

        def act_part():
            nonlocal act_prev
            if self.serial and self.ar_off_chain and k > 0:
                self.ev_upd[(k - 1) % 2].wait(self.s_env)      # ADAM(actor)_{k-1} ran on the side stream
            with torch.cuda.stream(self.s_env):
                if first:                                  # reset!(env): this step starts from the initial condition
                    y_in.copy_(env.y0)
                    s_in.copy_(self.state0)
                    act_prev = self.azero
                # actor forward on all B*A columns + exploration noise (Philox, counter on the device) + clamp: one launch
                L.check(lib.pdec_set_stream(self.actor.handle, self._sp_env))
                L.check(lib.pdec_policy_act_rng_dev(self.actor.handle, L.ptr(s_in), self.cols, float(pol.act_noise),
                                                    float(pol.act_limit), 1, self.noise_seed, L.ptr(act)))
                L.check(lib.pdec_set_stream(self.actor.handle, self._sp_upd))
                if not self.serial and self.LAG >= 2:
                    self.ev_act[k % 2].record(self.s_env)
            if self.drain_between:
                torch.cuda.synchronize()

        j = k - self.LAG
        # one stream + the collective off the chain: act_k needs ADAM(actor)_{k-1}, which waits for the all-reduce on the side
        # stream -- so the critic half of update_k, which needs neither, goes first and hides it (two streams: the critic half is
        # on the other stream anyway)
        late_env = self.serial and self.ar_off_chain and j >= self._first_tick and not self.use_replay
        if not late_env:
            act_part()

        def env_part():
            with torch.cuda.stream(self.s_env):
                L.check(lib.pdec_env_set_terminal_out(env.handle, L.ptr(term)))
                if self.rpart is not None:
                    L.check(lib.pdec_env_set_reward_partials_out(env.handle, L.ptr(self.rpart[k % 3]), None))
                L.check(lib.pdec_env_step(env.handle, L.ptr(y_in), L.ptr(act), L.ptr(act_prev), L.ptr(s_in), L.ptr(y_out),
                                          L.ptr(self.pbuf), L.ptr(s_out), L.ptr(rew), L.ptr(flags)))
                if last:
                    term.fill_(1.0)                    # time-out: done = time >= te -> terminal transition
                if self.pre_rbar and self.rpart is None:
                    L.check(lib.pdec_reward_mean(env.handle, L.ptr(rew), self.cols, L.ptr(self.rbar[k % 3])))
                if self.use_replay:
                    # a host function with step-dependent ring positions: re-evaluated, not replayed verbatim (_Lib.note)
                    lib.note(self._replay_push, self.ev_push[k % 2], s_in, act, rew, term, s_out, first, last)
                if not self.serial and self.LAG < 2:
                    # LAG = 1: update_{k+1} trains on the transition env_k is producing, so the event it waits on is
                    # recorded behind the whole env branch, not behind the acting kernel
                    self.ev_act[k % 2].record(self.s_env)
                if self._mid_waits:
                    self.ev_env[k % 2].record(self.s_env)
            if self.drain_between:                     # kernel-timing pass: nothing of the env branch overlaps the update
                torch.cuda.synchronize()

        batch = None
        if j >= self._first_tick:
            if self.use_replay:
                with torch.cuda.stream(self.s_upd):
                    lib.note(self._replay_sample, self.ev_push[(k - 1) % 2] if (not self.serial and k > 0) else None)
                batch = self._batch_cur
            else:
                batch = dict(state=self.sring[j % PERIOD].view(self.cols, self.ns), action=self.aring[j % 3].view(self.cols, self.na),
                             reward=self.rring[j % 3].view(self.cols), terminal=self.tring[j % 3].view(self.cols),
                             next_state=self.sring[(j + 1) % PERIOD].view(self.cols, self.ns))
        # (inside a captured chunk the extra fork / join edge of the kick costs more than it gives: 202 vs 148 us per step)
        kick = self.kick_env_after_critic and batch is not None and not self.serial and not capturing
        if not kick and not late_env:
            env_part()

        def between_halves():
            if late_env:
                act_part()
                env_part()
            # Beside the critic pass the PDE step makes almost no progress and then collides with the whole actor pass;
            # released when the critic half (pass + reduction) is done it runs beside the actor pass, the second
            # reduction and the head of the next critic pass instead (r02i, same box: 152 -> 130 us per control step)
            if kick:
                if use_stop:
                    L.check(lib.pdec_mlp_flush_stop_event(pol.behavior_critic.model.handle))     # no-op when consumed
                else:
                    self.ev_mid.record(self.s_upd)
                self.ev_mid.wait(self.s_env)
                env_part()
            if self.act_in_place and not self.serial:
                self.ev_act[k % 2].wait(self.s_upd)
            if self.ar_off_chain and k > 0:
                # the actor pass reads the actor that ADAM(actor)_{k-1} on the side stream has written (done long ago: it ran
                # beside the critic half that has just ended) ...
                self.ev_upd[(k - 1) % 2].wait(self.s_upd)
                if self._mid_waits:
                    # ... and the wait for the batch of update_{k+1} -- env_{k-1}, done since the middle of this critic pass --
                    # sits here too, instead of at the top of step k + 1.  Why: every cross-stream wait is a barrier packet of
                    # 5 - 7 us on the update chain even when it is satisfied (measured: the step with either wait removed),
                    # two adjacent ones cost less than two apart (116.5 - 117.7 against 118 - 119 us per step), a single
                    # wait on act_k's event (which implies both) or on a join event made on the side stream stalls on events
                    # that complete just before they are needed (123 / 125 us) -- HISTORY.md 6.1
                    self.ev_env[(k - 1) % 2].wait(self.s_upd)
                    self._mid_wait_prev = True

        # eager steps on the fused 3-layer path: the two events the env stream waits for ride on the reduction launches' own
        # dispatch packets (pdec_mlp_set_stop_event) instead of being recorded as packets behind them
        use_stop = self.stop_events and batch is not None and not capturing and not self.serial
        if use_stop:
            if kick:
                L.check(lib.pdec_mlp_set_stop_event(pol.behavior_critic.model.handle, self.ev_mid.h))
            L.check(lib.pdec_mlp_set_stop_event(pol.behavior_actor.model.handle, self.ev_upd[k % 2].h))
        with torch.cuda.stream(self.s_upd):
            if batch is not None:
                if self.rpart is not None:
                    L.check(lib.pdec_ddpg_set_reward_partials(pol.behavior_critic.model.handle, L.ptr(self.rpart[j % 3]), self.n_rpart))
                elif self.pre_rbar:
                    L.check(lib.pdec_ddpg_set_reward_mean(pol.behavior_critic.model.handle, L.ptr(self.rbar[j % 3])))
                if self.ar_off_chain:
                    self._update_ar_off_chain(k, batch, between_halves, use_stop)
                elif kick or (self.act_in_place and not self.serial):
                    pol.update(batch, before_actor_half=between_halves)
                else:
                    pol.update(batch)
            if self.ar_off_chain and batch is not None:
                pass                                    # (the completion event was recorded on the side stream)
            elif use_stop:
                L.check(lib.pdec_mlp_flush_stop_event(pol.behavior_actor.model.handle))          # no-op when consumed
            elif not self.serial:
                self.ev_upd[k % 2].record(self.s_upd)
        if chunk_last and not self.serial:
            self.ev_upd[k % 2].wait(self.s_env)    # join: the graph ends on the env stream

    def _update_ar_off_chain(self, k, batch, between_halves, use_stop):
        """update_k of a data-parallel run with the collective off the update chain (module docstring).  On the update stream:
        critic half, actor pass, slab reduction (its completion = `ev_red`, riding on the reduction launch where the fused path
        allows); on the side stream behind that event: all-reduce of the flat actor gradient, then the ADAM launch -- issued
        through the actor's handles while they are set to the side stream -- whose completion is `ev_upd[k % 2]`, the event the
        next acting kernel (env stream) and the next actor pass (update stream) wait for."""
        pol, lib, L = self.policy, self.lib, _lib
        A, At = pol.behavior_actor.model, pol.target_actor.model
        pol.update_critic_half(batch)
        between_halves()
        red_on_launch = use_stop and lib.pdec_mlp_set_reduce_event(A.handle, self.ev_red.h) == 0
        pol.actor_grads(batch)                                      # (records a reduce event its launches did not carry)
        if not red_on_launch:
            self.ev_red.record(self.s_upd)
        self.ev_red.wait(self.s_ar)
        pol.reducer.all_reduce(A, stream=self.s_ar)
        L.check(lib.pdec_set_stream(A.handle, self._sp_ar))
        L.check(lib.pdec_set_stream(At.handle, self._sp_ar))
        try:
            pol.apply_actor()
            if use_stop:
                L.check(lib.pdec_mlp_flush_stop_event(A.handle))    # no-op when the ADAM launch carried ev_upd[k % 2]
            else:
                self.ev_upd[k % 2].record(self.s_ar)
        finally:
            L.check(lib.pdec_set_stream(A.handle, self._sp_upd))
            L.check(lib.pdec_set_stream(At.handle, self._sp_upd))

    _capturing = False
    _first_tick = 0
    _mid_wait_prev = False    # update_{k-1} waited for env_{k-2} between its halves (so step k needs no wait at its top)
    drain_between = False

    def _first_last(self, k):
        """is step k the first / the last of an episode"""
        e = k - self.ep_start
        if self.E <= 0:
            return e == 0, False
        return e % self.E == 0, e % self.E == self.E - 1

    # ------------------------------------------------------------------ device replay route (row F1)
    def _replay_push(self, ev_done, s_in, act, rew, term, s_out, first, last):
        """the stage pushes of step k (src/PDEagent.jl:237-314).  They run on the ENV stream (launched through the env's
        handle), behind the env step / the time-out fill that produce their inputs; `ev_push[k % 2]` marks them done and
        the update that samples the replay next waits for it (see _issue).  (launched through the critic's
        handle they ran on the update stream, unordered against env_k, and copied stale rewards / flags.)"""
        tr, lib, h = self.agent.trajectory, self.lib, self.env.handle
        cap1 = tr.capacity + tr.stride
        ns, na = tr.state.shape[1], tr.action.shape[1]
        dtc = _lib.dtype_code(s_in.dtype)

        def push_sa(s, a):
            _lib.check(lib.pdec_replay_push_sa(h, _lib.ptr(tr.state), _lib.ptr(tr.action), cap1, ns, na, tr.n_sa % cap1,
                                               _lib.ptr(s.view(self.cols, self.ns)),
                                               None if a is None else _lib.ptr(a.view(self.cols, self.na)), self.cols, dtc))
            tr.n_sa += self.cols

        if self._samp_pending:
            self.ev_samp.wait(self.s_env)
            self._samp_pending = False
        if first and len(tr) > 0 and tr.n_sa > tr.n_rt:
            tr.pop_sa(tr.stride)                                    # PRE_EPISODE: the dummy (s, a) of the last episode
        push_sa(s_in, act)                                          # PRE_ACT
        # POST_ACT: reward and the per-column terminal flags the env step (and the time-out) produced -- two
        # one-column traces of equal capacity, pushed by the same two-trace kernel as (s, a)
        _lib.check(lib.pdec_replay_push_sa(h, _lib.ptr(tr.reward), _lib.ptr(tr.terminal), tr.capacity, 1, 1,
                                           tr.n_rt % tr.capacity, _lib.ptr(rew.view(self.cols)),
                                           _lib.ptr(term.view(self.cols)), self.cols, _lib.dtype_code(rew.dtype)))
        tr.n_rt += self.cols
        if last:
            push_sa(s_out, None)                                    # POST_EPISODE dummy
        if not self.serial:
            ev_done.record(self.s_env)

    def _replay_sample(self, ev_pushed):
        """the sampled batch of this step's update (pde_sample / pde_fetch!, src/PDEagent.jl:317-340) into the trajectory's
        fixed batch buffers -> self._batch_cur (None while the replay holds less than one step).  Re-evaluated at every step,
        recorded or not: the sample count, the ring fill and the Philox offset are host counters."""
        tr, pol = self.agent.trajectory, self.policy
        self._batch_cur = None
        if ev_pushed is not None:
            # the host counters the sample is drawn against already include env_{k-1}'s pushes: wait for them
            # (the update stream has otherwise only waited for act_{k-1}, which precedes them on the env stream)
            ev_pushed.wait(self.s_upd)
        if len(tr) <= tr.stride:
            return
        self._batch_cur = tr.sample_device(pol._sample_seed, pol._sample_off, self.cols, reuse=True)
        pol._sample_off += (self.cols + 3) // 4
        if not self.serial:
            # ... and env_k's pushes overwrite the oldest rows of a full ring: they wait for this sample
            self.ev_samp.record(self.s_upd)
            self._samp_pending = True

    # ------------------------------------------------------------------ graphs
    def capture(self):
        """record the chunk graphs (needs >= 3 warm-up steps behind it so that every lazily created buffer exists and
        the update has transitions to train on); advances the run by len(chunks) periods of eager-equivalent steps"""
        self._check_key()
        if not self.use_graphs or self._captured:
            return
        if self.E > 0:
            assert self.E > 2 * PERIOD, "episodes must be longer than two graph periods"
            self.chunks = tuple(c for c in self.chunks if c <= self.E - 2)
        while self.tick - self.LAG < self._first_tick + 2:      # lazily created buffers / first-use uploads happen eagerly
            self._eager()
        self._key = self._scalar_key()
        for c in self.chunks:
            for pos in range(PERIOD):
                if not self._reachable(c, pos):
                    # e.g. chunk 24 with 26-step episodes: the interior offsets 1 .. E-1-c never meet this ring phase
                    # (the search below would issue eager steps for ever); run() falls back to a smaller chunk
                    continue
                while not (self.tick % PERIOD == pos and self._interior(self.tick, c)):
                    self._eager()
                self._sync_streams_for_graph()
                _lib.check(self.lib.pdec_capture_begin(self.env.handle))
                self._capturing = True
                try:
                    for i in range(c):
                        self._issue(self.tick + i, chunk_first=(i == 0), chunk_last=(i == c - 1))
                finally:
                    self._capturing = False
                    h = _lib.Handle()
                    _lib.check(self.lib.pdec_capture_end(self.env.handle, C.byref(h)))
                self.graphs[(c, pos)] = h
                # the capture recorded the steps without running them: replay once so that the run really advances
                self._launch(h)
                self.tick += c
        self._captured = True

    def _reachable(self, c, pos):
        """is there a future tick at ring phase `pos` from which c consecutive steps are interior to an episode?  The
        pattern of (tick mod 6, episode offset) repeats after lcm(E, 6) <= 6 E ticks."""
        if self.E <= 0:
            return True
        t0 = max(self.tick, self._first_tick + self.LAG)
        return any(t % PERIOD == pos and self._interior(t, c) for t in range(t0, t0 + PERIOD * self.E + PERIOD))

    def _scalar_key(self):
        """the per-step scalars that a recorded step / a captured graph holds as frozen launch arguments"""
        pol = self.policy
        return (float(pol.act_noise), float(pol.act_limit), float(pol.behavior_actor.optimizer.eta),
                float(pol.behavior_critic.optimizer.eta), float(pol.y), float(pol.rho_effective), int(bool(pol.quirk)),
                bool(self.kick_env_after_critic), int(self.noise_seed))

    def _check_key(self):
        """a noise / learning-rate schedule (or load_agent restoring act_noise) changed a frozen scalar: drop the recorded
        steps and the graphs -- the run continues eagerly with the new values until capture() is called again"""
        key = self._scalar_key()
        if key != self._key:
            self._progs = {}
            if self.graphs:
                for h in self.graphs.values():
                    self.lib.pdec_destroy(h)
                self.graphs = {}
            self._captured = False
            self._key = key

    def _interior(self, k, n):
        """steps k .. k+n-1 are neither the first nor the last of an episode (those run eagerly)"""
        if k - self.LAG < self._first_tick:
            return False
        e = k - self.ep_start
        if self.E <= 0:
            return e >= 1
        e %= self.E
        return e >= 1 and e + n <= self.E - 1

    def _eager(self):
        """issue step `tick` call by call.  An interior step (not the first / last of an episode) at ring phase
        tick mod 6 makes exactly the library calls, with exactly the arguments, of every other interior step at that
        phase (the property the HIP graphs rest on), so the calls of the first such step are recorded and later ones
        replay the list -- ~20 raw C calls instead of the Python layers of PDEenv / policy / torch stream contexts"""
        k = self.tick
        fast = (self.fast_eager and not self.drain_between and not self._after_graph and k > 0
                and self._interior(k, 1) and self._interior(k - 1, 1))
        prog = self._progs.get(k % PERIOD) if fast else None
        if prog is not None:
            for f, a in prog:
                rc = f(*a)
                if rc:
                    _lib.check(rc)
        elif fast:
            calls = []
            self.lib.record_into(calls)
            try:
                self._issue(k)
            finally:
                self.lib.record_into(None)
            self._progs[k % PERIOD] = calls
        else:
            self._issue(k)
        self.tick += 1
        self.n_eager_steps += 1

    def _sync_streams_for_graph(self):
        """a graph runs wholly on the env stream: order the eager update still queued on the update stream before it"""
        if not self.serial and not self._after_graph and self.tick > 0:
            self.ev_upd[(self.tick - 1) % 2].wait(self.s_env)

    def _launch(self, h):
        _lib.check(self.lib.pdec_graph_launch(h, self._sp_env))
        if not self.serial:
            self.ev_graph.record(self.s_env)
        self._after_graph = True

    def run(self, n):
        """issue n control steps (asynchronous: returns when they are enqueued)"""
        n = int(n)
        self._check_key()
        while n > 0:
            k = self.tick
            done = False
            if self._captured:
                for c in self.chunks:
                    if c <= n and (c, k % PERIOD) in self.graphs and self._interior(k, c):
                        self._sync_streams_for_graph()
                        self._launch(self.graphs[(c, k % PERIOD)])
                        self.tick += c
                        n -= c
                        self.n_graph_launches += 1
                        done = True
                        break
            if not done:
                self._eager()
                n -= 1

    def step(self):
        self.run(1)

    def sync(self):
        self.s_env.synchronize()
        self.s_upd.synchronize()
        if self.s_ar is not None:
            self.s_ar.synchronize()

    def close(self):
        if getattr(self, "simd_sharing", False):
            self.lib.pdec_env_set_simd_sharing(self.env.handle, 0, None)
            self.simd_sharing = False
        for h in self.graphs.values():
            self.lib.pdec_destroy(h)
        self.graphs = {}
        self._captured = False
