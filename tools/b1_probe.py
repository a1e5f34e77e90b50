#!/usr/bin/env python3
"""Diagnostic: speed of the reference-shaped single-trajectory run loop (KS22, B = 1, 2-layer nets, batch_size 3,
update_loops 20) through the host mirror; optional cProfile of the host side."""
import cProfile
import importlib
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
which = next((a for a in sys.argv[1:] if not a.startswith("-")), "ks22")
if which == "ks22":
    setup, step_label = pkg.KSSetup.KS22(), b"ks_env_step"
elif which == "kseg":
    setup, step_label = pkg.KellerSegelSetup(), b"kseg_env_step"
else:
    setup, step_label = pkg.FluidSetup(nx=128), b"fluid_k1"
steps_per_ep = int(round((setup.te - setup.t0) / setup.dt)) + 1
two = "--two-streams" in sys.argv         # update beside the env step (run(): automatic with two explicit streams)
s_env, s_upd = (torch.cuda.Stream(), torch.cuda.Stream()) if two else (None, None)
env = pkg.PDEenv(setup, B=1, dtype=torch.float64, stream=s_env)
agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(0), stream=s_upd)
hook = pkg.PDEhook(min_best_episode=1, use_random_init=True, collect_bestDF="--bestdf" in sys.argv)
pkg.run(agent, env, pkg.StopAfterEpisodeWithMinSteps(60), hook)          # warm-up
nsteps = 2000 if which == "ks22" else (4000 if which == "kseg" else 120)
torch.cuda.synchronize()
n0 = len(hook.rewards)
pr = cProfile.Profile() if "--profile" in sys.argv else None
t0 = time.perf_counter()
if pr:
    pr.enable()
dev = None if "--stage-loop" not in sys.argv else False      # --stage-loop: the per-step host loop (rounds 1 - 5)
rt0 = agent.trajectory.n_rt
pkg.run(agent, env, pkg.StopAfterEpisodeWithMinSteps(nsteps), hook, device_episodes=dev)
if pr:
    pr.disable()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
eps = len(hook.rewards) - n0
ns_done = (agent.trajectory.n_rt - rt0) // agent.trajectory.stride
print(f"{which}: episodes {eps}, {ns_done} steps, {ns_done / dt:.0f} env-steps/s ({dt / max(1, ns_done) * 1e3:.2f} ms/step)")
if pr:
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
import ctypes as C
L = pkg._lib
cm = agent.policy.behavior_critic.model
L.check(cm.lib.pdec_prof_reset(cm.handle)); L.check(cm.lib.pdec_prof_enable(cm.handle, 1))
L.check(env.lib.pdec_prof_reset(env.handle)); L.check(env.lib.pdec_prof_enable(env.handle, 1))
pkg.run(agent, env, pkg.StopAfterEpisodeWithMinSteps(100), hook)
torch.cuda.synchronize()
for h, lab in ((cm.handle, b"ddpg_small"), (env.handle, step_label)):
    ms, n = C.c_double(), C.c_int()
    L.check(cm.lib.pdec_prof_get(h, lab, C.byref(ms), C.byref(n)))
    print(lab.decode(), f"{ms.value * 1e3:.1f} us/launch (events, incl. ~10 us event overhead), {n.value} launches")
