#!/usr/bin/env python3
"""Diagnostic: are two identical pipeline runs bit-identical (eager/eager, graph/graph, eager/graph)?"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("distributedconvrl-pde-control_amd")

def masked_streams():
    """two streams on DISJOINT halves of the CUs (hipExtStreamCreateWithCUMask): kernels of the env stream and of the update
    stream then never share a CU"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    out = []
    for half in (0, 1):
        mask = (C.c_uint32 * 8)(*([0xFFFFFFFF] * 4 + [0] * 4 if half == 0 else [0] * 4 + [0xFFFFFFFF] * 4))
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask)
        assert rc == 0, rc
        out.append(torch.cuda.ExternalStream(st.value))
    return out

def make(use_graphs, B=64, E=17):
    setup = pkg.KSSetup.bench_C2(256)
    s_env, s_upd = masked_streams() if os.environ.get("CUMASK") == "1" else (torch.cuda.Stream(), torch.cuda.Stream())
    y0 = setup.generate_random_init(np.random.default_rng(0), B) * 0.15
    env = pkg.PDEenv(setup, B=B, dtype=torch.float32, y0=y0, stream=s_env, autoreset=False)
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, stream=s_upd, start_steps=-1,
                             noise_seed=7, trajectory_length=1)
    agent.policy.act_noise = 0.3
    torch.cuda.synchronize()
    return pkg.TrainPipeline(env, agent, lag=2, episode_steps=E, stream_env=s_env, stream_upd=s_upd, use_graphs=use_graphs,
                             chunks=(6, 1), noise_seed=99)

def run(p, n, sync_each=False):
    if p.use_graphs:
        p.run(5); p.capture()
    while p.tick < n:
        p.run(1)
        if sync_each:
            torch.cuda.synchronize()
    p.sync()
    return [x.copy() for nm in ("behavior_actor", "behavior_critic", "target_actor", "target_critic") for x in getattr(p.policy, nm).model.params()] + [p.y.cpu().numpy()]

def same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))

def firstdiff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if not np.array_equal(x, y):
            return i, float(np.abs(x - y).max())
    return None

if __name__ == '__main__':
  n = 40
  ref = run(make(False), n, sync_each=True)
  for name, mk, se in (("eager", lambda: make(False), False), ("eager2", lambda: make(False), False), ("graph", lambda: make(True), False),
                       ("eager_synced", lambda: make(False), True)):
      r = run(mk(), n, se)
      print(f"PDEC_SPLIT={os.environ.get('PDEC_SPLIT')} {name:14s} == synced eager reference: {same(ref, r)}  first diff {firstdiff(ref, r)}")
