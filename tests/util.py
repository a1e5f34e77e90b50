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
