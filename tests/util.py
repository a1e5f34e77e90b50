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
