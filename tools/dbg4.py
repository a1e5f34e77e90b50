import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
import test_gpu_agent as T
from debug_graph_eq import snap, diff   # noqa
pe = T._make_pipeline(pkg, False)
for n in (1, 2, 3, 5, 10, 20, 40, 60):
    pe.run(n - pe.tick)
    s = snap(pe)
    bad = [k for k, v in s.items() if not (torch.isfinite(v).all().item() if isinstance(v, torch.Tensor) else np.isfinite(v).all())]
    print(n, "non-finite:", bad or "none", "losses", pe.policy.losses())
