#!/usr/bin/env python3
"""First episodes of many independently seeded training runs (tests/util.py: train) -- how wide the distribution of learning
curves is that the reference's single saved run is one draw from.
    python tools/train_seed_sweep.py kseg|ks22|fluid8 n_seeds n_loops [frozen=1]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
from util import train  # noqa: E402
from train_curve_probe import make  # noqa: E402


def main():
    which, n, loops = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    frozen = (sys.argv[4] if len(sys.argv) > 4 else "1") != "0"
    setup, kw, ref = make(which)
    kw["loops"] = loops
    np.set_printoptions(linewidth=220, precision=2, suppress=True)
    print("reference:", np.asarray(ref)[:4 * loops + 8])
    s_env, s_upd = torch.cuda.Stream(), torch.cuda.Stream()
    for seed in range(n):
        env = pkg.PDEenv(setup, B=1, dtype=torch.float64, stream=s_env)
        agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(500 + seed), noise_seed=1500 + seed, stream=s_upd,
                                 quirk_frozen_targets=frozen)
        hook = pkg.PDEhook(min_best_episode=1, use_random_init=True, init_seed=2500 + seed, init_rng=np.random.default_rng(seed))
        t = time.time()
        train(pkg, agent, env, hook, **kw)
        torch.cuda.synchronize()
        print(f"seed {seed:2d} ({time.time() - t:.1f} s):", np.asarray(hook.rewards), flush=True)


if __name__ == "__main__":
    main()
