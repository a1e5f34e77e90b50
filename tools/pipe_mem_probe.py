import importlib, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
setup = pkg.KSSetup.bench_C2(256, memory_size=2)
B = 8
s_env, s_upd = pkg.make_streams((-1, 0))
env = pkg.PDEenv(setup, B=B, dtype=torch.float32, stream=s_env, autoreset=False)
agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, stream=s_upd, start_steps=-1, trajectory_length=1)
pipe = pkg.TrainPipeline(env, agent, lag=2, episode_steps=17, stream_env=s_env, stream_upd=s_upd, use_graphs=False, noise_seed=5)
pipe.run(40); pipe.sync()
print("finite", bool(torch.isfinite(pipe.y).all()), agent.policy.losses(), tuple(env.state.shape), tuple(env.action.shape))
