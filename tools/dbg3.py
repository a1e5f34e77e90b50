import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
import test_gpu_agent as T
from debug_graph_eq import snap, diff   # noqa

def load(stream, n):
    a = torch.randn(2048, 2048, device="cuda")
    with torch.cuda.stream(stream):
        for _ in range(n):
            a = a @ a * 1e-3
    return a

def run_graph(noise):
    pg = T._make_pipeline(pkg, True)
    ls = torch.cuda.Stream()
    pg.run(5)
    if noise: load(ls, 40)
    pg.capture()
    if noise: load(ls, 200)
    for n in (1, 7, 20, 32):
        pg.run(n)
    torch.cuda.synchronize()
    return snap(pg), pg.tick

def run_eager(noise, upto):
    pe = T._make_pipeline(pkg, False)
    ls = torch.cuda.Stream()
    if noise: load(ls, 200)
    pe.run(upto)
    torch.cuda.synchronize()
    return snap(pe)


tick = int(sys.argv[1])
e0 = run_eager(False, tick)
e1 = run_eager(True, tick)
print(tick, "eager first-load vs alone", diff(e0, e1))
