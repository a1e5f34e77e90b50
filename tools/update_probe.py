"""Diagnostic: the DDPG update (critic pass, finish, actor pass, finish) of a config's networks ALONE on the device, Bu columns of
random data: per-kernel durations from the library's event pairs (pdec_prof_enable) and the wall time of N back-to-back updates.
usage: python tools/update_probe.py [C4|C5|C2|C3] [Bu]"""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
setup, B = {"C4": (lambda: pkg.KellerSegel2DSetup(nx=256, ny=256), 128), "C5": (lambda: pkg.FluidSetup(nx=512, sensors_per_axis=16, variance=0.04), 16),
            "C2": (lambda: pkg.KSSetup.bench_C2(256), 512), "C3": (lambda: pkg.KSSetup.bench_C2(1024), 512)}[cfg]
setup = setup()
ns, A = setup.state_shape
Bu = int(sys.argv[2]) if len(sys.argv) > 2 else B * A
st = torch.cuda.Stream()
agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), stream=st, max_update_cols=Bu, quirk_frozen_targets=True)
pol = agent.policy
g = torch.Generator().manual_seed(0)
batch = dict(state=torch.randn(Bu, ns, generator=g).cuda(), action=(torch.rand(Bu, 1, generator=g) * 2 - 1).cuda(),
             reward=-torch.rand(Bu, generator=g).cuda(), terminal=torch.zeros(Bu).cuda(), next_state=torch.randn(Bu, ns, generator=g).cuda())
L, lib = pkg._lib, pkg._lib.load()
with torch.cuda.stream(st):
    for _ in range(5):
        pol.update(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        pol.update(batch)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e6
    hs = [m.model.handle for m in (pol.behavior_actor, pol.behavior_critic)]
    for h in hs:
        L.check(lib.pdec_prof_reset(h)); L.check(lib.pdec_prof_enable(h, 1))
    for _ in range(10):
        pol.update(batch)
    torch.cuda.synchronize()
    out = {}
    for lab in ("ddpg2_critic_fused", "ddpg2_actor_fused", "fused2_finish", "ddpg_critic_fused", "ddpg_actor_fused", "fused_finish", "ddpg_rmean"):
        for h in hs:
            ms, k = C.c_double(), C.c_int()
            L.check(lib.pdec_prof_get(h, lab.encode(), C.byref(ms), C.byref(k)))
            if k.value:
                out[lab] = round(ms.value * 1e3, 1)
print(cfg, "Bu", Bu, "dims", pol.behavior_actor.model.dims, pol.behavior_critic.model.dims, "update wall us", round(wall, 1), out)
