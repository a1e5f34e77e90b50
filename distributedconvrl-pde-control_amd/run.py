"""The RL.jl run loop the reference relies on (`run(agent, env, stop_condition, hook)`), with
the stage order replicated in-tree by scripts/Fluid/setup/FluidSetup.jl:436-519, and the
reference's stop conditions (src/StopCondition.jl:6-40)."""
from .agent import (PRE_EXPERIMENT_STAGE, PRE_EPISODE_STAGE, PRE_ACT_STAGE, POST_ACT_STAGE, POST_EPISODE_STAGE,
                    POST_EXPERIMENT_STAGE)


class StopAfterEpisode:
    def __init__(self, episode):
        self.episode, self.cur = episode, 0

    def __call__(self, agent, env):
        if env.is_terminated():
            self.cur += 1
        return self.cur >= self.episode


class StopAfterEpisodeWithMinSteps:
    """src/StopCondition.jl:6-40: stop at the first episode end after >= `step` steps"""

    def __init__(self, step):
        self.step, self.cur = step, 1

    def __call__(self, agent, env):
        stop = self.cur >= self.step and env.is_terminated()     # :31
        self.cur += 1
        return stop


def device_episodes_ok(agent, env, stop_condition, hook):
    """can `run` issue whole episodes without reading the environment's flags back after every step (_run_device_episodes)?
    The reference's own shape: ONE trajectory (any environment kind: every output of a step goes to a per-step slot), the
    small-batch update with minibatches drawn on the device, the reference's zero start policy, a stop condition that fires at
    episode ends."""
    from .agent import Agent, CircularArraySARTTrajectory, CustomDDPGPolicy, ZeroPolicy
    from .hook import PDEhook
    if not isinstance(agent, Agent) or not isinstance(getattr(agent, "policy", None), CustomDDPGPolicy) or type(hook) is not PDEhook:
        return False
    pol, tr, setup = agent.policy, agent.trajectory, env.setup
    if type(stop_condition) not in (StopAfterEpisode, StopAfterEpisodeWithMinSteps):
        return False
    if env.B != 1 or env.autoreset or getattr(setup, "memory_size", 0):
        return False
    if getattr(env, "stream", None) is None or getattr(tr, "stream", None) is None or type(tr) is not CircularArraySARTTrajectory:
        return False
    if getattr(tr, "_h", None) is None or pol.sampling != "device" or not pol.small_update_ok() or pol.memory_size:
        return False
    if not (type(pol.start_policy) is ZeroPolicy or pol.start_steps <= 0) or hook.log_trajectory != 0:
        return False
    return tr.stride <= 256 and agent.after_push is None


def run(agent, env, stop_condition, hook, overlap=None, device_episodes=None):
    """RL.jl's `run(agent, env, stop_condition, hook)`: the stage order of every control step is the reference's
    (action = agent(env); PRE_ACT: push + update; env(action); POST_ACT: push r, t).

    overlap (default: automatic): when the environment and the agent's networks were created on two DIFFERENT explicit
    streams (`PDEenv(..., stream=s_env)`, `create_agent(..., stream=s_upd)`), the update of a control step and its env step
    run side by side -- they are independent in the reference's order too: the update samples what was pushed BEFORE this
    step's action acts, and the env step needs the action only.  Two events per step carry the true dependencies: the env
    step waits for the acting kernel and the PRE_ACT push (which read env.state) but not for the update; the POST_ACT push and
    the next acting kernel wait for the env step.  Same kernels, same arguments, same order per stream: results are bit-identical
    to the one-stream loop.  (For the reference-shaped single-trajectory loops the 20-update launch and the PDE step are of
    similar length: KS22 4.8 k -> 7+ k env-steps/s.)

    device_episodes (default: automatic, whenever device_episodes_ok): the control steps of a whole episode are ENQUEUED
    without waiting for any of them -- the one thing the host needs from the device per step, `is_terminated(env)`, is decided
    on the device (a halt flag, pdec_set_episode_halt) and read back once per episode.  Same kernels, same arguments, same
    order: bit-identical to the stage loop (tests/test_gpu_training.py)."""
    s_env, s_upd = getattr(env, "stream", None), getattr(getattr(agent, "trajectory", None), "stream", None)
    if device_episodes is None:
        device_episodes = device_episodes_ok(agent, env, stop_condition, hook)
    elif device_episodes and not device_episodes_ok(agent, env, stop_condition, hook):
        raise ValueError("run(device_episodes=True): this agent / environment / stop condition / hook needs the stage loop "
                         "(see device_episodes_ok)")
    if device_episodes:
        return _run_device_episodes(agent, env, stop_condition, hook, s_env, s_upd)
    two = s_env is not None and s_upd is not None and s_env.cuda_stream != s_upd.cuda_stream
    if overlap is None:
        overlap = two
    if overlap and not two:
        raise ValueError("run(overlap=True) needs the environment and the agent on two explicit streams")
    if not overlap:
        if two:                                # two streams without the event protocol would race: order them every stage
            return _run_two_streams_serial(agent, env, stop_condition, hook, s_env, s_upd)
        return _run_plain(agent, env, stop_condition, hook)
    import torch
    ev_act, ev_env = torch.cuda.Event(), torch.cuda.Event()

    def release_env_step():                    # between the PRE_ACT push and the update (Agent.after_push)
        ev_act.record(s_upd)
        s_env.wait_event(ev_act)

    def join():                                # episode boundaries: both streams see everything the other has done
        s_upd.wait_stream(s_env)
        s_env.wait_stream(s_upd)

    prev_hook, agent.after_push = getattr(agent, "after_push", None), release_env_step
    try:
        hook(PRE_EXPERIMENT_STAGE, agent, env)
        agent(PRE_EXPERIMENT_STAGE, env)
        is_stop = False
        while not is_stop:
            join()
            env.reset()
            join()
            agent(PRE_EPISODE_STAGE, env)
            hook(PRE_EPISODE_STAGE, agent, env)
            join()
            while not env.is_terminated():
                action = agent(env)            # s_upd (behind the update of the previous step; waited for the env step below)
                agent(PRE_ACT_STAGE, env, action)      # s_upd: push (s, a) | release_env_step | update
                hook(PRE_ACT_STAGE, agent, env)
                env(action)                    # s_env, beside the update
                ev_env.record(s_env)
                s_upd.wait_event(ev_env)       # POST_ACT push and the next acting kernel read what the step wrote
                agent(POST_ACT_STAGE, env)
                hook(POST_ACT_STAGE, agent, env)
                if stop_condition(agent, env):
                    is_stop = True
                    break
            if env.is_terminated():
                join()
                agent(POST_EPISODE_STAGE, env)
                hook(POST_EPISODE_STAGE, agent, env)
        join()
        hook(POST_EXPERIMENT_STAGE, agent, env)
    finally:
        agent.after_push = prev_hook
    return hook


class _EpisodeLogs:
    """per-step output slots of one episode: every launch of step t writes into slot t (t + 1 for what the NEXT step reads),
    so a step issued behind the end of the episode overwrites nothing that is still needed"""

    def __init__(self, env, T):
        import torch
        kw = dict(dtype=env.dtype, device=env.device)
        self.T = T
        self.y = torch.zeros((T + 1,) + env._yshape, **kw)
        self.state = torch.zeros((T + 1,) + env._sshape, **kw)
        self.action = torch.zeros((T + 1,) + env._ashape, **kw)       # slot 0 = action0, slot t + 1 = the action of step t
        self.p = torch.zeros((T,) + env._pshape, **kw)
        self.reward = torch.zeros((T, env.B, env.setup.reward_len), **kw)
        self.done = torch.zeros(T, dtype=torch.int32, device=env.device)
        self.halt = torch.zeros(1, dtype=torch.int32, device=env.device)


def _episode_steps(env):
    """control steps until time >= te, with the floating-point sum the step loop makes (50 x 0.1 != 5.0)"""
    t, n = 0.0, 0
    while True:
        t += env.dt
        n += 1
        if t >= env.te or n > 100000:
            return n


def _launch_sync_ok(env):
    """the fused fp64 KS step of one trajectory with 192 / 240 / 600 cells: the launch that honours pdec_set_launch_sync"""
    import torch
    s = env.setup
    return (getattr(s, "integrator", None) == "cnab2" and getattr(s, "nx", 0) in (192, 240, 600) and env.B == 1
            and env.dtype == torch.float64 and not getattr(s, "memory_size", 0))


def _step_glue(lib, pol, tr, env, logs, t, cols, A, acting, pending_rt, done_event=None):
    """[POST_ACT push of step t - 1] + agent(env) + PRE_ACT push of step t as one launch (pdec_step_glue); False = not served,
    nothing enqueued, no counter moved"""
    import ctypes as C

    from . import _lib
    if getattr(tr, "_h", None) is None or getattr(pol, "_glue_off", False):
        return False
    m = pol.behavior_actor.model
    P, na, ns = _lib.ptr, m.dims[-1], tr.state.shape[1]
    a_t = logs.action[t + 1]
    served = C.c_int(0)
    cap1 = tr.capacity + tr.stride
    n_rt = cols if pending_rt else 0
    _lib.check(lib.pdec_step_glue(
        m.handle, tr._h, _lib.dtype_code(env.dtype),
        P(logs.reward[t - 1].view(-1)) if pending_rt else None, P(logs.done[t - 1:t]) if pending_rt else None, int(A), 0,
        P(tr.reward), P(tr.terminal), tr.capacity, tr.n_rt % tr.capacity, n_rt,
        1 if acting else 2, P(logs.state[t]), cols, float(pol.act_noise), float(pol.act_limit), pol._noise_seed, pol._noise_off,
        P(a_t), P(tr.state), P(tr.action), cap1, tr.n_sa % cap1, cols, done_event.h if done_event is not None else _lib.Handle(0),
        C.byref(served)))
    if not served.value:
        pol._glue_off = True          # (the answer depends on shapes and streams only)
        return False
    tr.n_rt += n_rt
    tr.n_sa += cols
    if acting:
        pol._noise_off += (cols * na + 3) // 4
    return True


def _fast_issue(lib, pol, tr, env, agent, logs, T, cols, A, sync, seq):
    """The step loop of _run_device_episodes for the case where every hand-over is inside the kernels (glue launch served, fused
    fp64 KS step, small-batch update with device-side sampling): the SAME library calls with the SAME arguments as the general
    loop below, issued through the raw C functions from addresses computed once -- the general loop spends ~45 us of host time per
    step (25 pointer look-ups, five wrapped calls) where the GPU needs 57.  Returns issue(marks) or None when the case is not
    this one."""
    import ctypes as C

    from . import _lib
    if not (pol.small_update_ok() and pol.sampling == "device" and getattr(tr, "_h", None) is not None):
        return None
    raw = lib._c
    f_glue, f_sync, f_upd, f_env = raw.pdec_step_glue, raw.pdec_set_launch_sync, raw.pdec_ddpg_update_small_rng, raw.pdec_env_step
    check = _lib.check
    Am, Cm, Atm, Ctm = (pol.behavior_actor.model, pol.behavior_critic.model, pol.target_actor.model, pol.target_critic.model)
    hA, hC, hAt, hCt, henv, htr = Am.handle, Cm.handle, Atm.handle, Ctm.handle, env.handle, tr._h
    na = Am.dims[-1]
    dcode = _lib.dtype_code(env.dtype)

    def base(tns):                 # address of slot 0 and bytes per slot of a [slots, ...] log
        return tns.data_ptr(), tns[0].numel() * tns.element_size()
    (aS, sS), (aA, sA), (aY, sY), (aP, sP), (aR, sR) = base(logs.state), base(logs.action), base(logs.y), base(logs.p), base(logs.reward)
    aD = logs.done.data_ptr()
    pTS, pTA, pTR, pTT = tr.state.data_ptr(), tr.action.data_ptr(), tr.reward.data_ptr(), tr.terminal.data_ptr()
    cap, cap1, stride = tr.capacity, tr.capacity + tr.stride, tr.stride
    f0, f1 = sync.data_ptr(), sync.data_ptr() + 8
    noise, limit = float(pol.act_noise), float(pol.act_limit)
    loops, Bu = int(pol.update_loops), int(pol.batch_size)
    gamma, rho, quirk = float(pol.y), pol.rho_effective, int(pol.quirk)
    eta_a, eta_c = float(pol.behavior_actor.optimizer.eta), float(pol.behavior_critic.optimizer.eta)
    pL = pol._losses.data_ptr()
    d_noise, d_sample = (cols * na + 3) // 4, (loops * Bu + 3) // 4
    served = C.c_int(0)
    pserved = C.byref(served)
    after, freq, start_steps = pol.update_after * stride, pol.update_freq, pol.start_steps

    def issue(marks):
        n_sa, n_rt, ustep, noff, soff, q = tr.n_sa, tr.n_rt, pol.update_step, pol._noise_off, pol._sample_off, seq[0]
        seed_n, seed_s = pol._noise_seed, pol._sample_seed
        for t in range(T):
            ustep += 1
            q += 1
            acting = ustep > start_steps
            pend = t > 0
            rc = f_sync(hA, f1 if pend else None, q - 1, f0, q)
            rc = rc or f_glue(hA, htr, dcode, (aR + (t - 1) * sR) if pend else None, (aD + (t - 1) * 4) if pend else None, A, 0,
                              pTR, pTT, cap, n_rt % cap, cols if pend else 0, 1 if acting else 2, aS + t * sS, cols, noise, limit,
                              seed_n, noff, aA + (t + 1) * sA, pTS, pTA, cap1, n_sa % cap1, cols, 0, pserved)
            if rc or not served.value:
                check(rc)
                raise RuntimeError("run(device_episodes): pdec_step_glue stopped serving the loop it served at its start")
            if pend:
                n_rt += cols
            n_sa += cols
            if acting:
                noff += d_noise
            if min(n_rt, cap) > after and ustep % freq == 0:                   # Agent._maybe_update, src/PDEagent.jl:354-355
                rc = f_upd(hA, hC, hAt, hCt, pTS, pTA, pTR, pTT, loops, Bu, seed_s, soff, min(n_rt, cap), n_rt, cap, stride, gamma,
                           rho, quirk, eta_a, eta_c, pL)
                soff += d_sample
            rc = rc or f_sync(henv, f0, q, f1, q)
            rc = rc or f_env(henv, aY + t * sY, aA + (t + 1) * sA, aA + t * sA, aS + t * sS, aY + (t + 1) * sY, aP + t * sP,
                             aS + (t + 1) * sS, aR + t * sR, aD + t * 4)
            if rc:
                check(rc)
            marks.append((noff, soff))
        tr.n_sa, tr.n_rt, pol.update_step, pol._noise_off, pol._sample_off, seq[0] = n_sa, n_rt, ustep, noff, soff, q
    return issue


def _run_device_episodes(agent, env, stop_condition, hook, s_env, s_upd):
    """RL.jl's run loop with every episode issued in one go (module docstring of run / device_episodes_ok).  Stream protocol of
    the overlapped loop: acting kernel, PRE_ACT push and the update on the networks' stream, the env step on the environment's
    beside the update, POST_ACT push behind the env step.  What the host knows in advance -- ring positions, Philox offsets,
    the update trigger of src/PDEagent.jl:354-355 -- is passed as scalars, exactly the values the stage loop would pass; the
    one thing it does not know, whether a step blew the trajectory up, is the device's halt flag: the POST_ACT push of that
    step raises it and the pushes / updates of the later steps do nothing, all their other outputs go to per-step slots.  One
    read-back per episode tells how many steps counted; the host counters are set to that."""
    import ctypes as C

    import numpy as np
    import torch

    from . import _lib
    from .env import _on_stream

    pol, tr, lib = agent.policy, agent.trajectory, env.lib
    pol._glue_off = False                     # ask the library once per run whether it serves the one-launch glue
    two = s_env.cuda_stream != s_upd.cuda_stream
    ns, A = env.setup.state_shape
    cols, na = env.B * A, env._ashape[-1]
    T = _episode_steps(env)
    logs = getattr(env, "_episode_logs", None)
    if logs is None or logs.T != T:
        with _on_stream(s_env):
            logs = env._episode_logs = _EpisodeLogs(env, T)
    from .pipeline import _Event
    ev_act, ev_env = _Event(lib), _Event(lib)      # device-scope events: the hand-offs are on every step's chain
    P = _lib.ptr
    # device-side hand-overs (see the step loop): flags [glue, env step] and the sequence number of the step they count
    sync, seq = None, [0]
    if two and _launch_sync_ok(env) and getattr(tr, "_h", None) is not None:
        ok = C.c_int(0)
        _lib.check(lib.pdec_step_glue_served(pol.behavior_actor.model.handle, tr._h, _lib.dtype_code(env.dtype), 1, cols, cols,
                                             C.byref(ok)))
        if ok.value:      # ... and the two streams must sit on different hardware queues (a 20-ms bounded probe, once per run)
            _lib.check(lib.pdec_streams_run_side_by_side(C.c_void_p(s_env.cuda_stream), C.c_void_p(s_upd.cuda_stream), C.byref(ok)))
        if ok.value:
            with _on_stream(s_env):
                sync = torch.zeros(2, dtype=torch.int64, device=env.device)
            nto = C.c_int(0)
            _lib.check(lib.pdec_launch_sync_timeouts(C.byref(nto)))
            timeouts0 = [nto.value]
    np_dt = np.float64 if env.dtype == torch.float64 else np.float32

    fast = _fast_issue(lib, pol, tr, env, agent, logs, T, cols, A, sync, seq) if sync is not None else None

    def join():
        if two:
            s_upd.wait_stream(s_env)
            s_env.wait_stream(s_upd)

    class _Ended:                      # what the stop condition sees of the environment at a step
        def __init__(self, ended):
            self.ended = ended

        def is_terminated(self):
            return self.ended

    def episode():
        """enqueue the T control steps, read back once, settle the host state; returns True when the stop condition fired"""
        with _on_stream(s_env):
            logs.y[0].copy_(env.y)
            logs.state[0].copy_(env.state)
            logs.done.zero_()
            logs.halt.zero_()
        join()
        start = (pol.update_step, tr.n_sa, tr.n_rt, pol._noise_off, pol._sample_off)
        marks = []                                        # per step: (noise draws so far, sample offset so far)
        _lib.check(lib.pdec_set_episode_halt(tr._h, P(logs.halt)))
        try:
            with _on_stream(s_upd):
                pending_rt = False           # the POST_ACT push of step t - 1 rides on step t's glue launch
                if fast is not None:         # the same launches with the same arguments, issued from precomputed addresses
                    fast(marks)
                for t in (range(T) if fast is None else ()):
                    pol.update_step += 1
                    a_t = logs.action[t + 1]
                    acting = pol.update_step > pol.start_steps
                    # POST_ACT push of step t - 1 (:276-289), agent(env) (:175-209; ZeroPolicy: the zero action), PRE_ACT push
                    # (:254-274): ONE launch where the library serves it (pdec_step_glue), the three calls otherwise
                    # Hand-overs between the two streams: where the library serves both ends (the glue launch and the fused fp64
                    # KS step) they happen INSIDE the kernels -- the glue waits for env step t - 1's flag and raises its own,
                    # which env step t waits for (pdec_set_launch_sync; ~11 us per stream-level hop otherwise) --; else the
                    # event the env step waits for rides on the glue launch (no record packet on the update's stream)
                    if sync is not None:
                        seq[0] += 1
                        _lib.check(lib.pdec_set_launch_sync(pol.behavior_actor.model.handle,
                                                            P(sync[1:2]) if pending_rt else None, seq[0] - 1, P(sync[0:1]), seq[0]))
                    glued = _step_glue(lib, pol, tr, env, logs, t, cols, A, acting, pending_rt,
                                       ev_act if (two and sync is None) else None)
                    if not glued:
                        if pending_rt:
                            tr.push_rt_flags(logs.reward[t - 1].view(-1), logs.done[t - 1:t], A, False)
                        if acting:
                            pol.act_into(logs.state[t], cols, env.dtype, a_t.view(cols, na))
                        else:
                            a_t.zero_()
                        tr.push_sa(logs.state[t].view(cols, ns), a_t.view(cols, na))
                    if two and sync is None:
                        if not glued:
                            ev_act.record(s_upd)
                        ev_act.wait(s_env)
                    agent._maybe_update()                                      # :342-361
                    if sync is not None:
                        _lib.check(lib.pdec_set_launch_sync(env.handle, P(sync[0:1]), seq[0], P(sync[1:2]), seq[0]))
                    _lib.check(lib.pdec_env_step(env.handle, P(logs.y[t]), P(a_t), P(logs.action[t]), P(logs.state[t]), P(logs.y[t + 1]),
                                                 P(logs.p[t]), P(logs.state[t + 1]), P(logs.reward[t]), P(logs.done[t:t + 1])))
                    if two and (sync is None or t == T - 1):                   # (the last step's push is a launch of its own)
                        ev_env.record(s_env)
                        ev_env.wait(s_upd)
                    pending_rt = True
                    marks.append((pol._noise_off, pol._sample_off))
                if fast is not None:
                    ev_env.record(s_env)
                    ev_env.wait(s_upd)
                tr.push_rt_flags(logs.reward[T - 1].view(-1), logs.done[T - 1:T], A, True)      # the last step's: a time-out
                join()
                # the per-step episode reward of PDEhook (src/PDEhook.jl:51-63): mean over the actuators, summed over the steps
                means = logs.reward.reshape(T, -1).mean(dim=1)
                flags = logs.done.cpu().numpy()                                # the one read-back (waits for the whole episode)
        finally:
            _lib.check(lib.pdec_set_episode_halt(tr._h, None))
            if sync is not None:          # (an exception between a pdec_set_launch_sync and its launch must not leave it pending)
                lib.pdec_set_launch_sync(pol.behavior_actor.model.handle, None, 0, None, 0)
                lib.pdec_set_launch_sync(env.handle, None, 0, None, 0)
        if sync is not None:               # a hand-over that never came (0.3 s each) means the results are not the stage loop's
            nto = C.c_int(0)
            _lib.check(lib.pdec_launch_sync_timeouts(C.byref(nto)))
            if nto.value != timeouts0[0]:
                timeouts0[0] = nto.value
                raise RuntimeError("run(device_episodes): a device-side hand-over between the glue launch and the env step timed out")
        bad = np.flatnonzero(flags)
        n = int(bad[0]) + 1 if bad.size and bad[0] < T - 1 else T
        # ---- settle the host state at n executed steps
        pol.update_step = start[0] + n
        tr.n_sa, tr.n_rt = start[1] + n * cols, start[2] + n * cols
        pol._noise_off, pol._sample_off = marks[n - 1]
        env.y, env.state, env.prev_state = logs.y[n], logs.state[n], logs.state[n - 1]
        env.action, env._action_prev = logs.action[n], logs.action[n - 1]
        env._adopted.update((env.action.data_ptr(), env._action_prev.data_ptr()))          # views of the log: never written by the env
        env.p, env.reward, env._done_flags = logs.p[n - 1], logs.reward[n - 1], logs.done[n - 1:n]
        env.steps = n
        env.time = 0.0
        for _ in range(n):
            env.time += env.dt
        env._done_stale = True
        # ---- the hook's POST_ACT bookkeeping of the n steps (src/PDEhook.jl:51-63)
        m = means[:n].cpu().numpy().astype(np_dt)
        acc = m[0]
        for v in m[1:]:
            acc = np_dt(acc + v)
        hook.reward += float(acc)
        if hook.collect_bestDF:
            hook._rows_bulk = (list(range(1, n + 1)), logs.action[1:n + 1, 0], logs.p[:n, 0], logs.y[1:n + 1, 0], logs.reward[:n, 0])
        fired = False
        for i in range(n):
            fired = stop_condition(agent, _Ended(i == n - 1)) or fired
        return fired

    hook(PRE_EXPERIMENT_STAGE, agent, env)
    agent(PRE_EXPERIMENT_STAGE, env)
    is_stop = False
    while not is_stop:
        join()
        env.reset()
        join()
        agent(PRE_EPISODE_STAGE, env)
        hook(PRE_EPISODE_STAGE, agent, env)
        join()
        is_stop = episode()
        if env.is_terminated():
            join()
            agent(POST_EPISODE_STAGE, env)
            hook(POST_EPISODE_STAGE, agent, env)
    join()
    hook(POST_EXPERIMENT_STAGE, agent, env)
    return hook


def _run_two_streams_serial(agent, env, stop_condition, hook, s_env, s_upd):
    """the plain stage order for an environment and an agent that live on two streams: every stage sees all the other stream did"""
    def join():
        s_upd.wait_stream(s_env)
        s_env.wait_stream(s_upd)

    def staged(f, *a):
        join()
        out = f(*a)
        join()
        return out

    staged(hook, PRE_EXPERIMENT_STAGE, agent, env)
    staged(agent, PRE_EXPERIMENT_STAGE, env)
    is_stop = False
    while not is_stop:
        staged(env.reset)
        staged(agent, PRE_EPISODE_STAGE, env)
        staged(hook, PRE_EPISODE_STAGE, agent, env)
        while not env.is_terminated():
            action = staged(agent, env)
            staged(agent, PRE_ACT_STAGE, env, action)
            staged(hook, PRE_ACT_STAGE, agent, env)
            staged(env, action)
            staged(agent, POST_ACT_STAGE, env)
            staged(hook, POST_ACT_STAGE, agent, env)
            if stop_condition(agent, env):
                is_stop = True
                break
        if env.is_terminated():
            staged(agent, POST_EPISODE_STAGE, env)
            staged(hook, POST_EPISODE_STAGE, agent, env)
    staged(hook, POST_EXPERIMENT_STAGE, agent, env)
    return hook


def _run_plain(agent, env, stop_condition, hook):
    hook(PRE_EXPERIMENT_STAGE, agent, env)
    agent(PRE_EXPERIMENT_STAGE, env)
    is_stop = False
    while not is_stop:
        env.reset()
        agent(PRE_EPISODE_STAGE, env)
        hook(PRE_EPISODE_STAGE, agent, env)
        while not env.is_terminated():
            action = agent(env)
            agent(PRE_ACT_STAGE, env, action)
            hook(PRE_ACT_STAGE, agent, env)
            env(action)
            agent(POST_ACT_STAGE, env)
            hook(POST_ACT_STAGE, agent, env)
            if stop_condition(agent, env):
                is_stop = True
                break
        if env.is_terminated():
            agent(POST_EPISODE_STAGE, env)
            hook(POST_EPISODE_STAGE, agent, env)
    hook(POST_EXPERIMENT_STAGE, agent, env)
    return hook


def testrun(agent, env, steps=None, use_best=None, noise=False, log=True, seed=0):
    """One evaluation episode of the (best or current) actor without learning -- the rollout inside `testrun`
    (scripts/Fluid/setup/FluidSetup.jl:400-434) and `plot_heat` / `plotrun` (src/plotting.jl:4-60, :306-330), without their plots -- issued as ONE device-side rollout
    (pdec_rollout): env.reset(), then `steps` (default: one episode, te/dt + 1) control steps.  `use_best`: a
    PDEhook whose bestNNA should act instead of the policy's current actor.  Returns the rollout dictionary
    (reward_sum [B, A], done_step [B], and with log=True the per-step y / p / action / reward rows) plus
    `episode_reward` = the mean over actuators of the summed reward per trajectory, the figure PDEhook reports."""
    pol = agent.policy
    actor = (use_best.bestNNA if use_best is not None else pol.behavior_actor).model
    if actor.dtype != env.dtype or actor.max_cols < env.B * env.setup.state_shape[1]:
        actor = actor.clone(dtype=env.dtype, max_cols=env.B * env.setup.state_shape[1])
    env.reset()
    if steps is None:
        steps = int(round((env.te - env.t0) / env.dt)) + 1
    out = env.rollout(actor, steps, act_noise=pol.act_noise if noise else 0.0, act_limit=pol.act_limit, learning=bool(noise),
                      seed=seed, log=log)
    out["episode_reward"] = out["reward_sum"].mean(dim=1)
    return out
