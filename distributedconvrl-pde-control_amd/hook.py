"""PDEhook mirror (src/PDEhook.jl:8-103): reward bookkeeping, best-episode trajectory log
(`bestDF` rows timestep/action/p/y/reward), best-actor snapshot.  For B > 1 the episode
reward is the mean over the batch and the logged rows are those of trajectory 0."""
import copy

import numpy as np

from .agent import (PRE_EXPERIMENT_STAGE, PRE_EPISODE_STAGE, PRE_ACT_STAGE, POST_ACT_STAGE, POST_EPISODE_STAGE,
                    POST_EXPERIMENT_STAGE)


class PDEhook:
    def __init__(self, min_best_episode=0, use_random_init=False, collect_history=False, collect_NNA=True,
                 collect_bestDF=True, is_display_on_exit=False, error_detection=None, init_rng=None):
        self.rewards, self.rewards_compare = [], []
        self.reward, self.ep = 0.0, 1
        self.is_display_on_exit, self.use_random_init = is_display_on_exit, use_random_init
        self.collect_history, self.collect_NNA, self.collect_bestDF = collect_history, collect_NNA, collect_bestDF
        self.min_best_episode = min_best_episode
        self.bestNNA = self.currentNNA = None
        self.bestDF, self.currentDF = [], []
        self.bestreward, self.bestepisode = -1000000.0, 0
        self.history, self.errored_episodes = [], []
        self.error_detection = error_detection or (lambda y: False)
        self.init_rng = init_rng or np.random.default_rng(0)
        self._reward_dev, self._rows_dev = None, []

    def _flush(self, env):
        """bring the episode's device-side accumulators to the host (one synchronisation per episode)"""
        if self._reward_dev is not None:
            self.reward += float(self._reward_dev.item())
            self._reward_dev = None
        for steps, a, p, y, r in self._rows_dev:
            a, p, y = (t.cpu().numpy().astype(np.float64) for t in (a, p, y))
            if env.is_fluid:
                y, p = (y[..., 0] + 1j * y[..., 1]).T, (p[..., 0] + 1j * p[..., 1]).T
            elif y.ndim == 2:
                y = y.T
            self.currentDF.append(dict(timestep=steps, action=a.T.reshape(-1) if a.ndim == 2 else a.reshape(-1), p=p, y=y,
                                       reward=r.cpu().numpy().astype(np.float64)))
        self._rows_dev = []

    def __call__(self, stage, agent, env):
        if stage == PRE_EXPERIMENT_STAGE:                       # PDEhook.jl:35-40
            if self.collect_NNA and self.currentNNA is None:
                self.currentNNA = copy.deepcopy(agent.policy.behavior_actor)
                self.bestNNA = copy.deepcopy(agent.policy.behavior_actor)
        elif stage == PRE_EPISODE_STAGE:                        # :42-49
            if self.use_random_init:
                if hasattr(env.setup, "random_init_device"):   # initialiser kernel (row F4): no host field generation
                    env.y0 = env.setup.random_init_device(env, self.init_rng)
                else:
                    y0 = env.setup.generate_random_init(self.init_rng, env.B)
                    if y0.ndim == 3:
                        y0 = np.swapaxes(y0, 1, 2)
                    env.y0 = env._as_batch(y0, env._yshape)
                env.y.copy_(env.y0)
                env.state.copy_(env.featurize(env.y, env.state if env.setup.temporal_steps > 1 else None))
        elif stage == POST_ACT_STAGE:                           # :51-63
            # accumulated and logged ON THE DEVICE (row F4): no device->host copy or sync per control step; the host
            # values are materialised once per episode in POST_EPISODE_STAGE (send_to_host, PDEhook.jl:58-59)
            r = env.reward.mean()
            self._reward_dev = r if self._reward_dev is None else self._reward_dev + r
            if self.collect_bestDF:
                self._rows_dev.append((env.steps, env.action[0].clone(), env.p[0].clone(), env.y[0].clone(),
                                       env.reward[0].clone()))
        elif stage == POST_EPISODE_STAGE:                       # :65-97
            self._flush(env)
            if env.time >= env.te and self.ep >= self.min_best_episode:
                self.rewards_compare.append(self.reward)
                if self.collect_NNA and self.reward >= max(self.rewards_compare):
                    self.bestNNA.copyto(agent.policy.behavior_actor)
                    self.bestreward, self.bestepisode = self.reward, self.ep
                    if self.collect_bestDF:
                        self.bestDF = list(self.currentDF)
            if env.time < env.te and self.error_detection(env.y):
                self.errored_episodes.append(self.ep)
            if self.collect_history:
                self.history.append(self.currentDF)
            self.currentDF = []
            self.ep += 1
            self.rewards.append(self.reward)
            self.reward = 0.0
            if self.collect_NNA:
                self.currentNNA.copyto(agent.policy.behavior_actor)
        elif stage == POST_EXPERIMENT_STAGE:                    # :99-103
            if self.is_display_on_exit and self.rewards:
                print("Total reward per episode:", np.round(self.rewards, 4))
