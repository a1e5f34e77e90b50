import importlib, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
setup = pkg.KSSetup.KS22()
env = pkg.PDEenv(setup, B=1, dtype=torch.float64)
agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(0))
hook = pkg.PDEhook(min_best_episode=1, use_random_init=True, collect_bestDF=False)
t0 = time.perf_counter()
pkg.run(agent, env, pkg.StopAfterEpisodeWithMinSteps(200), hook)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = sum(1 for _ in range(1))
print("episodes", len(hook.rewards), "time", dt, "s ; steps/s ~", (len(hook.rewards) * 51) / dt, "rewards", hook.rewards[-3:])
