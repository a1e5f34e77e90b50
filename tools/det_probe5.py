#!/usr/bin/env python3
"""Diagnostic: two pipelines with the same seeds that differ only in TIMING / issue mechanics (SIDE_A / SIDE_B: environment of
each side, default: A reduces with the round-2 finish kernel, PDEC_FINISH_REF=1, whose summation tree is the same) must stay
bit-identical.  For n = 1 .. N fresh pairs run n steps in one
call; the first n at which any quantity differs is reported with the quantities that differ."""
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


BASE = {k: os.environ[k] for k in ("PDEC_FINISH_REF", "PDEC_STOP_EVENTS", "PDEC_KICK", "PDEC_SHARE", "PDEC_FAST_EAGER", "SYNC") if k in os.environ}
B, E = int(os.environ.get("B", "64")), int(os.environ.get("E", "23"))

def state(p):
    red = pkg.distributed.GradReducer()
    out = {"y": p.y.clone(), "critic_grad": red._view(p.policy.behavior_critic.model).clone(),
           "actor_grad": red._view(p.policy.behavior_actor.model).clone()}
    for k in range(3):
        out[f"action[{k}]"] = p.aring[k].clone(); out[f"reward[{k}]"] = p.rring[k].clone()
    for nm in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        out[nm] = torch.as_tensor(np.concatenate([x.ravel() for x in getattr(p.policy, nm).model.params()]))
    return out

SIDES = [os.environ.get("SIDE_A", "PDEC_FINISH_REF=1"), os.environ.get("SIDE_B", "")]
KEYS = ("PDEC_FINISH_REF", "PDEC_STOP_EVENTS", "PDEC_KICK", "PDEC_SHARE", "PDEC_FAST_EAGER", "SYNC")

def one(n, side):
    """side: 'NAME=value,...' environment of this pipeline; SYNC=1 drains the device after every step"""
    for k in KEYS:
        if k not in BASE:
            os.environ.pop(k, None)
        else:
            os.environ[k] = BASE[k]
    for kv in filter(None, SIDES[side].split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    p = make(False, B=B, E=E)
    if os.environ.get("SYNC") == "1":
        for _ in range(n):
            p.run(1); torch.cuda.synchronize()
    else:
        p.run(n)
    p.sync()
    s = state(p)
    p.close()
    return s

tag = f"CUMASK={os.environ.get('CUMASK')} B={B} base {BASE} A: {SIDES[0]} | B: {SIDES[1]}"
N = int(os.environ.get("N", "30"))
nbad = 0
for n in range(1, N + 1):
    a, b = one(n, 0), one(n, 1)
    bad = [(k, float((a[k].float() - b[k].float()).abs().max()), int((a[k] != b[k]).sum())) for k in a if not torch.equal(a[k], b[k])]
    if bad:
        nbad += 1
        if os.environ.get("SIG") == "1":          # compact signature: which quantities differ, in how many elements
            print(f"[sig] n = {n}: " + " ".join(f"{k}:{c}" for k, _, c in bad))
        elif nbad <= 2:
            print(f"[{tag}] n = {n}: {bad}")
print(f"[{tag}] {nbad} of {N} run lengths differ")
