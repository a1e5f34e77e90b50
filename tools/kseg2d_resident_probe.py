#!/usr/bin/env python3
"""What could a time-resident 2-D Keller-Segel RK4 kernel cost at best?  (HISTORY.md round 5.)
pdec_debug_kseg2d_probe runs the product's fp32 tile kernel with `reps` sub-steps per launch on the tile it holds in registers
(no halo refresh -- the instruction mix of a resident kernel without any exchange) on nb trajectories:
  * nb = 32 (512 tiles = one workgroup generation, two per CU), reps = 1: one sub-step launch without a tail;
  * reps = 8 / 32: the marginal cost of a sub-step that loads and stores nothing  -> (t(32) - t(8)) / 24;
  * nb = 128: four generations.
"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("distributedconvrl-pde-control_amd")


def main():
    setup = pkg.KellerSegel2DSetup()
    env = pkg.PDEenv(setup, B=128, dtype=torch.float32)
    lib = pkg._lib.load()
    us = C.c_double()
    res = {}
    # clocks up, and the product's own sub-step loop as the yardstick of this process (32 launches per call, whole batch per launch
    # with PDEC_KSEG2D_SPLIT=0, three parts on three streams otherwise)
    y = torch.ones((128, 256, 256, 2), dtype=torch.float32, device="cuda:0")
    p = torch.zeros((128, 256, 256), dtype=torch.float32, device="cuda:0")
    for _ in range(30):
        env.do_step(y, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        env.do_step(y, p)
    e1.record()
    torch.cuda.synchronize()
    print(f"product do_step (32 sub-step launches, PDEC_KSEG2D_SPLIT={os.environ.get('PDEC_KSEG2D_SPLIT', 'default')}): "
          f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us = {e0.elapsed_time(e1) / 20 / 32 * 1e3:.2f} us per sub-step", flush=True)
    for nb in (32, 64, 128):
        for reps in (1, 2, 8, 32):
            pkg._lib.check(lib.pdec_debug_kseg2d_probe(env.handle, nb, reps, 20 if reps < 32 else 8, C.byref(us)))
            res[(nb, reps)] = us.value
            print(f"nb={nb:4d} ({nb * 16:5d} tiles) reps={reps:3d}: {us.value:9.1f} us per launch, {us.value / reps:7.2f} us per sub-step", flush=True)
    for nb in (32, 64, 128):
        m = (res[(nb, 32)] - res[(nb, 8)]) / 24
        print(f"nb={nb}: marginal resident sub-step {m:.2f} us; load+store+launch of a one-sub-step launch = {res[(nb, 1)] - m:.2f} us; "
              f"32 resident sub-steps of {nb} trajectories >= {res[(nb, 32)]:.0f} us (+ halo exchange)")


if __name__ == "__main__":
    main()
