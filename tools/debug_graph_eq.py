#!/usr/bin/env python3
"""Diagnostic: where does a graph-replayed pipeline diverge from the eager one?"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
from test_gpu_agent import _make_pipeline


def snap(p):
    p.sync()
    d = dict(y=p.y.clone(), state=p.state.clone())
    for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        d[n] = np.concatenate([x.ravel() for x in getattr(p.policy, n).model.params()])
    return d


def diff(a, b):
    out = []
    for k in a:
        x, y = a[k], b[k]
        same = torch.equal(x, y) if isinstance(x, torch.Tensor) else np.array_equal(x, y)
        if not same:
            d = (x - y).abs().max().item() if isinstance(x, torch.Tensor) else np.abs(x - y).max()
            out.append(f"{k}:{d:.2e}")
    return out or ["identical"]


for serial in ((False,) if __name__ == '__main__' else ()):
    if os.environ.get("DBG_SKIP_EE") != "1":
        print("== eager vs eager")
        a, b = _make_pipeline(pkg, False), _make_pipeline(pkg, False)
        for n in (5, 10, 20):
            a.run(n); b.run(n)
            print(a.tick, diff(snap(a), snap(b)))
    print("== graph vs eager, step by step after capture")
    pe, pg = _make_pipeline(pkg, False), _make_pipeline(pkg, True)
    pg.run(5)
    import ctypes as C
    t0 = pg.tick
    # capture step by step, comparing after each graph
    orig_launch = pg._launch
    def launch(h):
        orig_launch(h)
    pg.capture()
    for key, h in pg.graphs.items():
        n = C.c_int()
        pkg._lib.check(pg.lib.pdec_graph_num_nodes(h, C.byref(n)))
        print("graph", key, "nodes", n.value)
    pe.run(pg.tick)
    if os.environ.get("DBG_SYNC", "1") == "1":
        print("after capture", pg.tick, diff(snap(pe), snap(pg)))
    import os
    seq = [int(x) for x in os.environ.get("DBG_SEQ", "").split(",") if x]
    do_sync = os.environ.get("DBG_SYNC", "1") == "1"
    if seq:
        for n in seq:
            pe.run(n); pg.run(n)
            if do_sync:
                print(pg.tick, "e=%d" % ((pg.tick - pg.ep_start) % pg.E), diff(snap(pe), snap(pg)))
        print("final", pg.tick, diff(snap(pe), snap(pg)))
        sys.exit(0)
    step = int(os.environ.get("DBG_STEP", "1"))
    for i in range(12):
        pe.run(step); pg.run(step)
        print(pg.tick, "e=%d" % ((pg.tick - pg.ep_start) % pg.E), "graphs", pg.n_graph_launches, "eager", pg.n_eager_steps, diff(snap(pe), snap(pg)))
