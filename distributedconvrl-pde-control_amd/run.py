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


def run(agent, env, stop_condition, hook, overlap=None):
    """RL.jl's `run(agent, env, stop_condition, hook)`: the stage order of every control step is the reference's
    (action = agent(env); PRE_ACT: push + update; env(action); POST_ACT: push r, t).

    overlap (default: automatic): when the environment and the agent's networks were created on two DIFFERENT explicit
    streams (`PDEenv(..., stream=s_env)`, `create_agent(..., stream=s_upd)`), the update of a control step and its env step
    run side by side -- they are independent in the reference's order too: the update samples what was pushed BEFORE this
    step's action acts, and the env step needs the action only.  Two events per step carry the true dependencies: the env
    step waits for the acting kernel and the PRE_ACT push (which read env.state) but not for the update; the POST_ACT push and
    the next acting kernel wait for the env step.  Same kernels, same arguments, same order per stream: results are bit-identical
    to the one-stream loop.  (For the reference-shaped single-trajectory loops the 20-update launch and the PDE step are of
    similar length: KS22 4.8 k -> 7+ k env-steps/s.)"""
    s_env, s_upd = getattr(env, "stream", None), getattr(getattr(agent, "trajectory", None), "stream", None)
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
