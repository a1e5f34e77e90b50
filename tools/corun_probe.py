#!/usr/bin/env python3
"""Diagnostic (not part of the product): can the KS env-step kernel share the GPU with the fused critic pass?
Times each alone and both concurrently on two streams, for several critic batch sizes."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
L = pkg._lib
lib = L.init(0)
dev = "cuda:0"
B = 512
setup = pkg.KSSetup.bench_C2(256)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
env = pkg.PDEenv(setup, B=B, dtype=torch.float32, device=dev, y0=setup.generate_random_init(np.random.default_rng(0), B) * 0.1, stream=s2)
agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, device=dev, stream=s1,
                         start_steps=-1, trajectory_length=1)
pol = agent.policy
act = torch.zeros(env._ashape, device=dev)


def run(n_env, n_upd, Bu):
    cols = Bu
    s = torch.randn(cols, 3, device=dev); a = torch.rand(cols, 1, device=dev); r = -torch.rand(cols, device=dev)
    t = torch.zeros(cols, device=dev); sn = torch.randn(cols, 3, device=dev)
    A, C, At, Ct = (pol.behavior_actor.model, pol.behavior_critic.model, pol.target_actor.model, pol.target_critic.model)
    losses = torch.zeros(2, device=dev)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        e[0].record(s1)
        for _ in range(n_upd):
            L.check(lib.pdec_ddpg_critic_grads(A.handle, C.handle, At.handle, Ct.handle, L.ptr(s), L.ptr(a), L.ptr(r), L.ptr(t),
                                               L.ptr(sn), cols, 0.99, 1, 1.0, None))
        e[1].record(s1)
    with torch.cuda.stream(s2):
        e[2].record(s2)
        for _ in range(n_env):
            env(act)
        e[3].record(s2)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tu = e[0].elapsed_time(e[1]) / max(n_upd, 1) * 1e3
    te = e[2].elapsed_time(e[3]) / max(n_env, 1) * 1e3
    return tu, te, wall * 1e6


for Bu in (32768, 16384, 8192):
    run(5, 5, Bu)
    print(f"Bu={Bu}: critic alone {run(0, 50, Bu)[0]:.1f} us/call | env alone {run(50, 0, Bu)[1]:.1f} us/step | "
          f"together: critic {run(50, 50, Bu)[0]:.1f}, env {run(50, 50, Bu)[1]:.1f} (us per call)")
