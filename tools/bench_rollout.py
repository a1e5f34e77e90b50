#!/usr/bin/env python3
"""Auxiliary bench (row F2): acting-only rollouts -- actor forward + noise + clamp + fused env step per control step,
no update -- issued (a) step by step from Python and (b) as ONE pdec_rollout call.  Prints one JSON line per batch size."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    pkg = importlib.import_module("distributedconvrl-pde-control_amd")
    L = pkg._lib
    T = 200
    cases = [("KS N=256 A=64", B, torch.float32) for B in (1, 16, 512)]
    cases += [("Keller-Segel 1-D (KellerSegelSetup)", B, dt) for B, dt in ((1, torch.float64), (64, torch.float64), (512, torch.float32))]
    for name, B, dtype in cases:
        if name.startswith("KS "):
            setup = pkg.KSSetup.bench_C2(256)
            y0 = setup.generate_random_init(np.random.default_rng(0), B) * 0.15
        else:
            setup = pkg.KellerSegelSetup()
            y0 = np.ascontiguousarray(np.swapaxes(setup.generate_random_init(np.random.default_rng(0), B), 1, 2))
        env = pkg.PDEenv(setup, B=B, dtype=dtype, y0=y0)
        agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=dtype, start_steps=-1)
        actor = agent.policy.behavior_actor.model
        cols = B * setup.n_actuators
        bufs = [torch.empty(env._ashape, dtype=dtype, device="cuda:0") for _ in range(2)]

        def loop(n):
            off = 0
            for t in range(n):
                a = bufs[t & 1]
                L.check(env.lib.pdec_policy_act_rng(actor.handle, L.ptr(env.state), cols, 0.3, 1.0, 1, 7, off, L.ptr(a)))
                off += (cols + 3) // 4
                env(a, adopt=True)

        loop(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(T)
        torch.cuda.synchronize()
        t_loop = time.perf_counter() - t0
        env.rollout(actor, 20, act_noise=0.3, learning=True, seed=7)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = env.rollout(actor, T, act_noise=0.3, learning=True, seed=7)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_roll = time.perf_counter() - t0
        print(json.dumps({"case": f"{name} acting-only rollout, B={B}, T={T}, {'fp32' if dtype == torch.float32 else 'fp64'}",
                          "python_loop_env_steps_per_s": B * T / t_loop, "python_loop_us_per_step": t_loop / T * 1e6,
                          "rollout_env_steps_per_s": B * T / t_roll, "rollout_us_per_step": t_roll / T * 1e6,
                          "rollout_host_issue_us_per_step": t_issue / T * 1e6,
                          "finite": bool(torch.isfinite(out["reward_sum"]).all().item())}))


if __name__ == "__main__":
    main()
