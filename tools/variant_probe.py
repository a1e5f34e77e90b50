import sys, os, json, copy, gc, argparse
sys.path.insert(0, os.getcwd())
import bench, torch
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-variants"]
# build args like main() does
ap_args = None
import types
def mk(**over):
    ns = types.SimpleNamespace(gpus=1, steps=None, warmup=None, batch=None, nx=None, no_cpu_baseline=True, config="C2", split_update=False,
        no_variants=True, deadline_s=1500.0, integrator="cnab2", no_overlap=False, overlap="auto", issue="auto", no_graph=False, replay=False,
        replay_steps=64, lag=2, dp_sync="policy", collective="auto", two_layer=False, episode_steps=51, cpu_seconds=12.0, repeats=0,
        launch_check=False, quick=True)
    for k, v in over.items(): setattr(ns, k, v)
    bench.fill_defaults(ns)
    return ns
bench.protect_stdout()
order = sys.argv  # unused
seq = os.environ.get("SEQ", "C4,KS,C4,C3,C4").split(",")
for s in seq:
    if s == "C4":
        o = bench.bench_aux(mk(config="C4", steps=int(os.environ.get("C4S","10")), warmup=int(os.environ.get("C4W","14"))))
    elif s == "KS":
        o = bench.bench_ks(mk(config="C2", steps=20, warmup=10))
    elif s.startswith("SLEEP"):
        import time
        time.sleep(float(s[5:]))
        continue
    elif s == "FD":
        o = bench.bench_ks(mk(config="C2", steps=20, warmup=10, integrator="rk4_fd"))
    elif s == "KSF":
        o = bench.bench_ks(mk(config="C2", steps=50, warmup=10, quick=False))
    elif s == "CPU":
        import importlib
        pkg = importlib.import_module("distributedconvrl-pde-control_amd")
        r = bench.cpu_baseline(pkg, pkg.KSSetup.bench_C2(256), 4.0)
        o = {"value": r["value"], "ms_per_step": 0.0, "kernels_ms_per_step": {}}
    elif s == "C5":
        o = bench.bench_aux(mk(config="C5", steps=1, warmup=1))
    elif s == "C3":
        a = mk(config="C3", steps=20, warmup=10); a.no_overlap = True
        o = bench.bench_ks(a)
    print(s, round(o["value"]), round(o["ms_per_step"], 4), {k: v for k, v in o["kernels_ms_per_step"].items() if k.startswith("kseg") or k.startswith("ddpg2")}, file=sys.stderr, flush=True)
    o = None
    gc.collect(); torch.cuda.synchronize()
    torch.cuda.empty_cache()
