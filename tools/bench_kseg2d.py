#!/usr/bin/env python3
"""Auxiliary bench (not the headline line of bench.py): the 2-D Keller-Segel env step of BASELINE.json configs[3]
(256 x 256, B = 128, RK4 with 32 sub-steps).  One JSON line per dtype: env-steps/s, the mean duration of one control
step's RK4 launches (HIP events through pdec_prof_*) and the HBM roofline fraction of the tile kernel against its
compulsory traffic (read y + p, write y per sub-step: 5 scalars per cell)."""
import argparse
import ctypes as C
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=256)
    ap.add_argument("--B", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtypes", default="f32,f64")
    ap.add_argument("--full", action="store_true", help="whole RL step: actor on all B*A columns, env step, one DDPG "
                    "update on the B*A fresh transitions (fp32; actor 36->20->1, critic 37->340->1 as in the 1-D script)")
    args = ap.parse_args()
    pkg = importlib.import_module("distributedconvrl-pde-control_amd")
    L = pkg._lib
    for name in args.dtypes.split(","):
        dt = torch.float32 if name == "f32" else torch.float64
        setup = pkg.KellerSegel2DSetup(nx=args.nx, ny=args.nx)
        rng = np.random.default_rng(0)
        y0 = np.moveaxis(setup.generate_random_init(rng, args.B), 1, -1)
        env = pkg.PDEenv(setup, B=args.B, dtype=dt, y0=np.ascontiguousarray(y0))
        act = torch.as_tensor(rng.uniform(-1, 1, env._ashape), dtype=dt, device="cuda:0")
        for _ in range(2):
            env(act)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            env(act)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / args.steps
        L.check(env.lib.pdec_prof_reset(env.handle))
        L.check(env.lib.pdec_prof_enable(env.handle, 1))
        for _ in range(3):
            env(act)
        torch.cuda.synchronize()
        ms, cnt = C.c_double(), C.c_int()
        L.check(env.lib.pdec_prof_get(env.handle, b"kseg2d_rk4", C.byref(ms), C.byref(cnt)))
        L.check(env.lib.pdec_prof_enable(env.handle, 0))
        ts = 4 if name == "f32" else 8
        cells = args.B * args.nx * args.nx
        alg = 5 * ts * cells * setup.oversampling            # per control step
        gbs = alg / (ms.value * 1e-3) / 1e9
        print(json.dumps({
            "case": f"kseg2d {args.nx}x{args.nx} B={args.B} {name} K={setup.oversampling} A={setup.n_actuators}",
            "env_steps_per_s": args.B / sec, "ms_per_control_step": sec * 1e3, "rk4_ms_per_control_step": ms.value,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                         "algorithmic_bytes_per_control_step": alg},
            "finite": bool(torch.isfinite(env.y).all().item()), "max_abs_y": float(env.y.abs().max().item())}))


def full(args):
    pkg = importlib.import_module("distributedconvrl-pde-control_amd")
    L = pkg._lib
    dt = torch.float32
    setup = pkg.KellerSegel2DSetup(nx=args.nx, ny=args.nx)
    rng = np.random.default_rng(0)
    y0 = np.ascontiguousarray(np.moveaxis(setup.generate_random_init(rng, args.B), 1, -1))
    env = pkg.PDEenv(setup, B=args.B, dtype=dt, y0=y0)
    agent = pkg.create_agent(setup=setup, B=args.B, rng=np.random.default_rng(1), dtype=dt, device="cuda:0",
                             start_steps=-1, noise_seed=7, trajectory_length=1)
    policy = agent.policy
    policy.act_noise = 0.3
    actor = policy.behavior_actor.model
    ns, A = setup.state_shape
    cols = args.B * A
    acts = [torch.empty((cols, 1), dtype=dt, device="cuda:0") for _ in range(2)]
    term = torch.zeros(cols, dtype=dt, device="cuda:0")
    env.set_terminal_out(term)
    off = [0]
    k = [0]
    EPISODE = 50

    def step():
        a = acts[k[0] % 2]
        k[0] += 1
        L.check(env.lib.pdec_policy_act_rng(actor.handle, L.ptr(env.state), cols, policy.act_noise, policy.act_limit, 1, 7,
                                            off[0], L.ptr(a)))
        off[0] += (cols + 3) // 4
        s_t = env.state
        env(a.view(env._ashape), adopt=True)
        end = k[0] % EPISODE == 0
        if end:
            term.fill_(1.0)             # time-out terminal (done = time >= te, src/PDEenv.jl:227)
        policy.update(dict(state=s_t.view(cols, ns), action=env.action.view(cols, 1), reward=env.reward.view(cols),
                           terminal=term, next_state=env.state.view(cols, ns)))
        if end:
            env.reset_episode()         # lock-stepped episodes: the chemotaxis model blows up in finite time under random forcing

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / args.steps
    al, cl = policy.losses()
    print(json.dumps({"case": f"kseg2d full RL step {args.nx}x{args.nx} B={args.B} f32 A={A} cols={cols} "
                              f"actor {policy.behavior_actor.model.dims} critic {policy.behavior_critic.model.dims}",
                      "env_steps_per_s": args.B / sec, "ms_per_step": sec * 1e3, "actor_loss": al, "critic_loss": cl,
                      "finite": bool(torch.isfinite(env.y).all().item())}))


if __name__ == "__main__":
    if "--full" in sys.argv:
        ap = argparse.ArgumentParser()
        ap.add_argument("--nx", type=int, default=256)
        ap.add_argument("--B", type=int, default=128)
        ap.add_argument("--steps", type=int, default=10)
        ap.add_argument("--full", action="store_true")
        full(ap.parse_args())
        sys.exit(0)
    main()
