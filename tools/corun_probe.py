#!/usr/bin/env python3
"""Diagnostic: the fused KS env step (SIMD-sharing form) on FIXED inputs on one stream, repeated, while another stream runs
the critic pass back to back.  Every repetition of the step must give the same bits."""
import ctypes as C
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
L = pkg._lib
B = int(os.environ.get("B", "512"))
setup = pkg.KSSetup.bench_C2(256)
s_env, s_upd = torch.cuda.Stream(), torch.cuda.Stream()
y0 = setup.generate_random_init(np.random.default_rng(0), B) * 0.15
env = pkg.PDEenv(setup, B=B, dtype=torch.float32, y0=y0, stream=s_env, autoreset=False)
share = os.environ.get("SHARE", "1") == "1"
env.set_simd_sharing(share)
agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, stream=s_upd, start_steps=-1,
                         noise_seed=7, trajectory_length=1)
pol = agent.policy
A_, Cn, At, Ct = (pol.behavior_actor.model, pol.behavior_critic.model, pol.target_actor.model, pol.target_critic.model)
cols = B * 64
g = torch.Generator(device="cuda").manual_seed(3)
s = torch.randn(cols, 3, device="cuda", generator=g); sn = torch.randn(cols, 3, device="cuda", generator=g)
a = torch.rand(cols, 1, device="cuda", generator=g) * 2 - 1; r = -torch.rand(cols, device="cuda", generator=g)
t = torch.zeros(cols, device="cuda")
act = torch.rand(env._ashape, device="cuda", generator=g) * 2 - 1
actp = torch.zeros_like(act)
y_in = env.y.clone(); st_in = env.state.clone()
outs = [torch.empty_like(y_in) for _ in range(2)]
p_out = torch.empty(env._pshape, device="cuda"); st_out = torch.empty_like(st_in)
rew = torch.empty((B, 64), device="cuda"); flags = torch.zeros(B, dtype=torch.int32, device="cuda")
Lz = torch.zeros(2, device="cuda")
torch.cuda.synchronize()
P = L.ptr

def step(out):
    with torch.cuda.stream(s_env):
        L.check(env.lib.pdec_env_step(env.handle, P(y_in), P(act), P(actp), P(st_in), P(out), P(p_out), P(st_out), P(rew), P(flags)))

def burn():
    which = os.environ.get("BURN", "critic")
    with torch.cuda.stream(s_upd):
        if which == "critic":
            L.check(A_.lib.pdec_ddpg_critic_grads(A_.handle, Cn.handle, At.handle, Ct.handle, P(s), P(a), P(r), P(t), P(sn), cols, 0.99, 1, 1.0,
                                                  C.c_void_p(Lz.data_ptr())))
        elif which == "actor":
            L.check(A_.lib.pdec_ddpg_actor_grads(A_.handle, Cn.handle, P(s), cols, 1.0, C.c_void_p(Lz.data_ptr() + 4)))

step(outs[0]); torch.cuda.synchronize()
ref = outs[0].clone()
bad = badg = 0
which = os.environ.get("BURN", "critic")
gview = None
if which != "none":
    gview = pkg.distributed.GradReducer()._view(Cn if which == "critic" else A_)
    burn(); torch.cuda.synchronize()
    gref = gview.clone()
n = int(os.environ.get("N", "600"))
for it in range(n):
    if os.environ.get("BURN", "critic") != "none":
        burn()
    step(outs[1])
    if it % int(os.environ.get('EVERY', '8')) == int(os.environ.get('EVERY', '8')) - 1:
        torch.cuda.synchronize()
        if gview is not None and not torch.equal(gview, gref):
            badg += 1
            if badg <= 3:
                d = (gview - gref).abs()
                print(f"   iteration {it}: gradient of the pass differs in {int((d > 0).sum())} of {d.numel()} elements, max {float(d.max()):.3e} (|g| max {float(gref.abs().max()):.3e})")
        if not torch.equal(outs[1], ref):
            bad += 1
            if bad <= 3:
                d = (outs[1] - ref).abs()
                rows = torch.nonzero(d.amax(dim=1)).flatten().tolist()
                print(f"   iteration {it}: {int((d > 0).sum())} differing cells, max {float(d.max()):.3e}, trajectories {rows[:8]}")
torch.cuda.synchronize()
print(f"SHARE={int(share)} BURN={os.environ.get('BURN', 'critic')}: {bad} of {n // int(os.environ.get('EVERY', '8'))} checked repetitions differ in the PDE fields, {badg} in the gradient of the pass")
