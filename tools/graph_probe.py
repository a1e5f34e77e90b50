#!/usr/bin/env python3
"""Diagnostic: does replaying the 4-launch DDPG update as a HIP graph shorten the inter-kernel gaps?  (Timing only: the
ADAM bias-correction powers are launch arguments, so a replayed graph reuses the captured ones.)"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
setup = pkg.KSSetup.bench_C2(256)
B = 512
cols = B * setup.n_actuators
s_upd = torch.cuda.Stream()
agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, start_steps=-1, stream=s_upd)
pol = agent.policy
g = torch.Generator(device="cuda").manual_seed(0)
batch = dict(state=torch.randn(cols, 3, device="cuda", generator=g), action=torch.rand(cols, 1, device="cuda", generator=g) * 2 - 1,
             reward=-torch.rand(cols, device="cuda", generator=g), terminal=torch.zeros(cols, device="cuda"),
             next_state=torch.randn(cols, 3, device="cuda", generator=g))
with torch.cuda.stream(s_upd):
    for _ in range(20):
        pol.update(batch)
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
with torch.cuda.stream(s_upd):
    for _ in range(n):
        pol.update(batch)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / n * 1e6
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s_upd):
    pol.update(batch)
torch.cuda.synchronize()
for _ in range(20):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    graph.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) / n * 1e6
print(f"update chain: eager {eager:.1f} us, graph replay {rep:.1f} us per update")
