"""Probe: does the fluid step gain from splitting the batch over two streams?  One PDEenv of B = 16 against two of B = 8 on two
streams (independent trajectories; kernels of different character -- K1 / K2 fp64-issue-bound, K3 HBM-bound -- may overlap)."""
import sys, os, time, importlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
n = int(os.environ.get("N", "512"))
setup = pkg.FluidSetup(nx=n, sensors_per_axis=16 if n >= 256 else 8, variance=0.04 if n >= 256 else 0.08)
def mk(B, stream):
    env = pkg.PDEenv(setup, B=B, dtype=torch.float64, device="cuda:0", autoreset=False, stream=stream)
    y0 = setup.random_init_device(env, np.random.default_rng(0)); env.set_y0(y0); env.reset()
    return env
def run(envs, steps=2):
    acts = [torch.zeros(e._ashape, dtype=torch.float64, device="cuda:0") for e in envs]
    for e, a in zip(envs, acts): e(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        for e, a in zip(envs, acts): e(a)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps
one = mk(16, torch.cuda.Stream())
t1 = run([one])
del one
a, b = mk(8, torch.cuda.Stream()), mk(8, torch.cuda.Stream())
t2 = run([a, b])
t8 = run([a])
print(f"n={n}: one env B=16: {t1*1e3:.1f} ms/step ({16/t1:.1f} env-steps/s); two envs B=8 on two streams: {t2*1e3:.1f} ms ({16/t2:.1f}); one env B=8 alone: {t8*1e3:.1f} ms ({8/t8:.1f})")
