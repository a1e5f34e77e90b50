import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLD, name))


def ks_pair(pkg, name):
    """(product setup, oracle config, golden) for a shipped KS experiment"""
    from oracle import ks
    g = load_golden(f"{name}_hook.npz")
    nx, Lx, stride, sigma = int(g["nx"]), float(g["Lx"]), int(g["sensor_stride"]), float(g["sigma"])
    mono = name.endswith("global")
    pos = np.arange(1, nx + 1, stride)
    setup = pkg.KSSetup(nx, Lx, pos, sigma_sensors=sigma, sigma_actuators=sigma, mono=mono)
    cfg = ks.KSConfig(nx, Lx, pos, sigma_sensors=sigma, sigma_actuators=sigma, mono=mono,
                      disturbance_in_step=not mono)
    return setup, cfg, g


def to_dev(a, dtype, device="cuda:0"):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=device)


def train(pkg, agent, env, hook, loops, no_steps, decay, use_random_init=True):
    """The `train()` of the reference's setup scripts (scripts/KS/setup/KSSetup.jl:304-319: 8 loops of >= 800 steps, act_noise
    x 0.2 per loop; scripts/Keller-Segel/setup/KellerSegelSetup.jl:390-406: 13 x 5000, x 0.6; scripts/Fluid/setup/
    FluidSetup.jl:541-556: x 0.6) -- a driver script (out of the product's scope), restated here as TEST
    infrastructure around the product's run() / agent / env / hook."""
    hook.use_random_init = use_random_init
    agent.policy.act_noise = env.setup.act_noise
    for _ in range(loops):
        pkg.run(agent, env, pkg.StopAfterEpisodeWithMinSteps(no_steps), hook)
        agent.policy.act_noise = agent.policy.act_noise * decay
        hook.rewards = [float(np.clip(r, -3000, 0)) for r in hook.rewards]
    return hook


def rlcore_wrap_shift(tr):
    """RLCore's CircularArraySARTTrajectory keeps capacity + 1 frames of state / action and `capacity` frames of reward /
    terminal, and pde_sample (src/PDEagent.jl:317-340) uses ONE index for all four traces.  Index i of a buffer that has seen G
    pushes and holds len = min(G, its capacity) frames is logical row G - len + i, so once the traces have wrapped, (s, a, s')
    come from rows (G_s - len_s) - (G_r - len_r) later than (r, t): A - 1 rows at the PRE_ACT update (the A state rows of this
    step are pushed, its rewards are not) -- the state and action of the NEXT control step of the neighbouring actuator.  Read off
    the reference's own KS200 buffer (tests/test_replay_golden.py).  The product keeps the four traces aligned; this function
    and `emulate_rlcore_wrap` below are STUDY code (effect of the misalignment on the learning curve: KS200's buffer wraps at
    episode 37, its reference curve relapses from episode 39) and live with the tests, not in the product."""
    cap = tr.capacity
    return (tr.n_sa - min(tr.n_sa, cap + 1)) - (tr.n_rt - min(tr.n_rt, cap))


def emulate_rlcore_wrap(pkg, agent):
    """make `agent` sample its minibatches like the reference's misaligned buffer: its trajectory becomes a subclass whose
    host-side pde_sample shifts the (s, a, s') slots by rlcore_wrap_shift, and the policy samples on the host"""
    import importlib
    base = importlib.import_module(pkg.__name__ + ".agent").CircularArraySARTTrajectory

    class RLCoreWrapTrajectory(base):
        def sample_slots(self, rng, batch_size):
            hi = len(self) - self.stride
            lg = max(0, self.n_rt - self.capacity) + rng.integers(0, hi, batch_size)
            ls = lg + rlcore_wrap_shift(self)
            cap1 = self.capacity + self.stride
            return ls % cap1, lg % self.capacity, (ls + self.stride) % cap1

        def sample_slots_many(self, rng, batch_size, loops):
            hi = len(self) - self.stride
            lg = max(0, self.n_rt - self.capacity) + rng.integers(0, hi, (loops, batch_size))
            ls = lg + rlcore_wrap_shift(self)
            cap1 = self.capacity + self.stride
            return np.stack([ls % cap1, lg % self.capacity, (ls + self.stride) % cap1])

    agent.trajectory.__class__ = RLCoreWrapTrajectory
    agent.policy.sampling = "host"
    return agent
