"""PDEhook mirror (src/PDEhook.jl:8-103): reward bookkeeping, best-episode trajectory log
(`bestDF` rows timestep/action/p/y/reward), best-actor snapshot.  For B > 1 the episode
reward is the mean over the batch and the logged rows are those of trajectory `log_trajectory`
(default 0; "best" = the trajectory with the highest episode return, chosen on the device at
the episode's end from rows logged for every trajectory)."""
import copy

import numpy as np

from .agent import (PRE_EXPERIMENT_STAGE, PRE_EPISODE_STAGE, PRE_ACT_STAGE, POST_ACT_STAGE, POST_EPISODE_STAGE,
                    POST_EXPERIMENT_STAGE)


class PDEhook:
    def __init__(self, min_best_episode=0, use_random_init=False, collect_history=False, collect_NNA=True,
                 collect_bestDF=True, is_display_on_exit=False, error_detection=None, init_rng=None, log_trajectory=0,
                 init_seed=0):
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
        self.log_trajectory = log_trajectory
        self.init_seed, self._init_off = int(init_seed), 0      # Philox stream of the device-side initialisers
        self._reward_dev, self._rows_dev, self._ret_dev = None, [], None
        self._rows_bulk = None       # (steps, action [n, ...], p, y, reward): a whole episode's rows in four device tensors

    def _flush(self, env):
        """bring the episode's device-side accumulators to the host (one synchronisation per episode)"""
        if self._reward_dev is not None:
            self.reward += float(self._reward_dev.item())
            self._reward_dev = None
        pick = None
        if self.log_trajectory == "best" and self._ret_dev is not None:
            pick = int(self._ret_dev.argmax().item())            # one scalar per episode crosses to the host
        self._ret_dev = None
        rows = self._rows_dev
        if self._rows_bulk is not None:      # run(device_episodes=True): one device->host copy per array instead of one per row
            steps_l, ab, pb, yb, rb = self._rows_bulk
            ab, pb, yb, rb = (t.cpu() for t in (ab, pb, yb, rb))
            rows = rows + [(st, ab[i], pb[i], yb[i], rb[i]) for i, st in enumerate(steps_l)]
            self._rows_bulk = None
        for steps, a, p, y, r in rows:
            if pick is not None:
                a, p, y, r = a[pick], p[pick], y[pick], r[pick]
            a, p, y = (t.cpu().numpy().astype(np.float64) for t in (a, p, y))
            if env.is_fluid:
                y, p = (y[..., 0] + 1j * y[..., 1]).T, (p[..., 0] + 1j * p[..., 1]).T
            elif y.ndim == 2:
                y = y.T
            self.currentDF.append(dict(timestep=steps, action=a.T.reshape(-1) if a.ndim == 2 else a.reshape(-1), p=p, y=y,
                                       reward=r.cpu().numpy().astype(np.float64)))
        self._rows_dev = []

    def __call__(self, stage, agent, env):
        # the hook's device ops (clones, accumulators, re-initialisation) are ordered with the env's kernels
        from .env import _on_stream
        with _on_stream(getattr(env, "stream", None)):
            self._call(stage, agent, env)

    def _call(self, stage, agent, env):
        if stage == PRE_EXPERIMENT_STAGE:                       # PDEhook.jl:35-40
            if self.collect_NNA and self.currentNNA is None:
                self.currentNNA = copy.deepcopy(agent.policy.behavior_actor)
                self.bestNNA = copy.deepcopy(agent.policy.behavior_actor)
        elif stage == PRE_EPISODE_STAGE:                        # :42-49
            if self.use_random_init:
                # generate_random_init() as an initialiser kernel (row F4): no host field generation, no upload.
                # Fluid: pdec_fluid_ic (the random vortex table is drawn on the host, 4 numbers per vortex);
                # KS / Keller-Segel: pdec_env_random_init (Philox stream (init_seed, offset) on the device)
                if getattr(env, "is_fluid", False) and hasattr(env.setup, "random_init_device"):
                    env.y0 = env.setup.random_init_device(env, self.init_rng)
                elif not getattr(env.setup, "is_kseg2d", False) and getattr(env.setup, "device_random_init", True):
                    y0 = env.y0 if env.y0.data_ptr() != env.y.data_ptr() else env.y0.clone()
                    self._init_off += env.random_init(self.init_seed, self._init_off, out=y0)
                    env.y0 = y0
                else:
                    y0 = env.setup.generate_random_init(self.init_rng, env.B)
                    if y0.ndim == 3:
                        y0 = np.swapaxes(y0, 1, 2)
                    env.y0 = env._as_batch(y0, env._yshape)
                env.y.copy_(env.y0)
                env.state.copy_(env.featurize(env.y, env.state if env.setup.temporal_steps > 1 else None))
                env._state0.copy_(env.state)      # what a per-trajectory reset (autoreset) restores
        elif stage == POST_ACT_STAGE:                           # :51-63
            # accumulated and logged ON THE DEVICE (row F4): no device->host copy or sync per control step; the host
            # values are materialised once per episode in POST_EPISODE_STAGE (send_to_host, PDEhook.jl:58-59)
            r = env.reward.mean()
            self._reward_dev = r if self._reward_dev is None else self._reward_dev + r
            if self.collect_bestDF:
                if self.log_trajectory == "best":      # rows of every trajectory stay on the device until the episode's end
                    rb = env.reward.mean(dim=1)
                    self._ret_dev = rb if self._ret_dev is None else self._ret_dev + rb
                    self._rows_dev.append((env.steps, env.action.clone(), env.p.clone(), env.y.clone(), env.reward.clone()))
                else:
                    b = int(self.log_trajectory)
                    self._rows_dev.append((env.steps, env.action[b].clone(), env.p[b].clone(), env.y[b].clone(),
                                           env.reward[b].clone()))
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
