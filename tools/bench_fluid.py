#!/usr/bin/env python3
"""Auxiliary bench (not the headline line of bench.py): the 2-D fluid path (src/fluid_rk4.jl rk4/rhs/advection +
FluidSetup.jl closures) at the reference's grid sizes and at BASELINE.json configs[4] (512 x 512 fp64, B = 64 over
4 GPUs = 16 per GPU).  Prints one JSON line per case: env-steps/s, per-kernel replay timings (HIP events through
pdec_prof_*), and the HBM roofline fraction of the three RHS kernels against their compulsory traffic."""
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
    ap.add_argument("--cases", default="128:16,256:16,512:16", help="nx:B list")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--substeps", type=int, default=0, help="override K (0 = floor(16 nx dt) as in FluidSetup.jl:47)")
    args = ap.parse_args()
    pkg = importlib.import_module("distributedconvrl-pde-control_amd")
    L = pkg._lib
    for case in args.cases.split(","):
        n, B = (int(v) for v in case.split(":"))
        setup = pkg.FluidSetup(nx=n, sensors_per_axis=16 if n >= 256 else 8, variance=0.04 if n >= 256 else 0.08,
                               oversampling=args.substeps or None)
        rng = np.random.default_rng(0)
        y0 = np.stack([setup.ic(3, rng)] * 1)
        y0 = np.repeat(y0, B, axis=0)
        env = pkg.PDEenv(setup, B=B, dtype=torch.float64, y0=y0)
        lib = env.lib
        act = torch.zeros(env._ashape, dtype=torch.float64, device="cuda:0")
        env(act)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            env(act)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        L.check(lib.pdec_prof_reset(env.handle))
        L.check(lib.pdec_prof_enable(env.handle, 4))
        z = torch.zeros_like(env.y)
        env.rhs(env.y, z)
        torch.cuda.synchronize()
        kern = {}
        for lab in ("fluid_k1", "fluid_k2", "fluid_k3"):
            ms, cnt = C.c_double(), C.c_int()
            L.check(lib.pdec_prof_get(env.handle, lab.encode(), C.byref(ms), C.byref(cnt)))
            kern[lab] = ms.value
        L.check(lib.pdec_prof_enable(env.handle, 0))
        p = 3 * n // 2
        z16 = 16
        # compulsory HBM bytes per trajectory and launch (complex fp64 = 16 B): K1 reads w (+ mirrored line) and writes
        # 2 (n+1) p; K2 reads that and writes n p; K3 reads n p + w + p^ (+ f, acc in RK4 stages) and writes rhs
        bytes_k = {"fluid_k1": (2 * n * n + 2 * (n + 1) * p) * z16, "fluid_k2": (2 * (n + 1) * p + n * p) * z16,
                   "fluid_k3": (n * p + 3 * n * n) * z16}
        roof = {k: {"ms": kern[k], "GB/s": B * bytes_k[k] / (kern[k] * 1e-3) / 1e9 if kern[k] else None,
                    "frac_of_8TBs": (B * bytes_k[k] / (kern[k] * 1e-3) / 8e12) if kern[k] else None} for k in kern}
        K = setup.oversampling
        rhs_ms = sum(kern.values())
        print(json.dumps({"metric": "env-steps/sec (2-D fluid rk4, fp64)", "value": B / dt, "unit": "env-steps/s",
                          "ms_per_step": dt * 1e3, "config": {"workload": f"fluid nx=ny={n} ifpad=1 K={K} B={B} fp64"},
                          "rhs_ms": rhs_ms, "rhs_share_of_step": 4 * K * rhs_ms / (dt * 1e3),
                          "algorithmic_GB_per_env_step_reference_unfused": 141.0 * (n / 512.0) ** 2 * (K / 163.0),
                          "kernels": roof}))
        env.close()


if __name__ == "__main__":
    main()
