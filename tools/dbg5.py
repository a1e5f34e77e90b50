import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
import test_gpu_agent as T
from debug_graph_eq import snap, diff   # noqa
sync_each = os.environ.get("SYNC_EACH") == "1"
a = T._make_pipeline(pkg, False)
junk = [torch.randn(1000 + 37 * i, 513, device="cuda") for i in range(7)]
x = torch.randn(2048, 2048, device="cuda"); x = x @ x
del junk
b = T._make_pipeline(pkg, False)
first = None
for k in range(1, 139):
    a.run(1); b.run(1)
    if sync_each or k in (1, 2, 3, 4, 5, 10, 20, 40, 80, 138):
        d = diff(snap(a), snap(b))
        if d != ["identical"] and first is None:
            first = k
            print("first difference at step", k, d)
print("final", diff(snap(a), snap(b)), "first", first)
