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


def run(agent, env, stop_condition, hook):
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
