import sys, os, time, importlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
n, B = 256, 128
setup = pkg.KellerSegel2DSetup(nx=n, ny=n)
rng = np.random.default_rng(0)
y0 = np.ascontiguousarray(np.moveaxis(setup.generate_random_init(rng, B), 1, -1))
for two in (True, False):
    s_env = torch.cuda.Stream(priority=-1)
    s_upd = torch.cuda.Stream() if two else s_env
    env = pkg.PDEenv(setup, B=B, dtype=torch.float32, device="cuda:0", y0=y0, stream=s_env, autoreset=False)
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, device="cuda:0", stream=s_upd,
                             start_steps=-1, noise_seed=1234, trajectory_length=1)
    agent.policy.act_noise = 0.3
    torch.cuda.synchronize()
    pipe = pkg.TrainPipeline(env, agent, lag=2, episode_steps=50, stream_env=s_env, stream_upd=s_upd, use_graphs=False, noise_seed=1234)
    pipe.run(6); torch.cuda.synchronize()
    t0 = time.perf_counter(); pipe.run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("two streams" if two else "one stream", "ms/step", dt / 40 * 1e3, "env-steps/s", B * 40 / dt, "finite", bool(torch.isfinite(pipe.y).all()), "kick", pipe.kick_env_after_critic, flush=True)
    pipe.close()
