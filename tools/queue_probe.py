"""Probe: how do torch stream pools map onto HIP's hardware queues?  For k = 0..7 dummy streams taken from the default pool before
the pipeline's two streams are created, time the C2 pipeline with the env stream from the high-priority pool and from the default
pool.  A collision (both streams on one hardware queue) shows as the serial time (~160 us per step instead of ~110)."""
import sys, os, time, importlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
setup = pkg.KSSetup.bench_C2(256)
B = 512
y0 = setup.generate_random_init(np.random.default_rng(0), B) * 0.15
def one(prio_env, prio_upd, skip_lo, skip_hi):
    keep = [torch.cuda.Stream() for _ in range(skip_lo)] + [torch.cuda.Stream(priority=-1) for _ in range(skip_hi)]
    s_env, s_upd = torch.cuda.Stream(priority=prio_env), torch.cuda.Stream(priority=prio_upd)
    env = pkg.PDEenv(setup, B=B, dtype=torch.float32, y0=y0, stream=s_env, autoreset=False)
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, stream=s_upd, start_steps=-1, noise_seed=7, trajectory_length=1)
    agent.policy.act_noise = 0.3
    torch.cuda.synchronize()
    p = pkg.TrainPipeline(env, agent, lag=2, episode_steps=51, stream_env=s_env, stream_upd=s_upd, use_graphs=False, noise_seed=99)
    p.run(30); torch.cuda.synchronize()
    t0 = time.perf_counter(); p.run(200); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    p.close()
    return dt * 1e6
for pe, pu in ((-1, 0), (0, 0), (-1, -1)):
    row = []
    for k in range(6):
        row.append(round(one(pe, pu, k if pu == 0 else 0, k if pu == -1 else 0), 1))
    print("env prio", pe, "upd prio", pu, "us/step after k extra streams of the update's pool:", row, flush=True)
