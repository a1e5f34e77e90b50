#!/usr/bin/env python3
"""Diagnostic: which quantity differs FIRST between an un-synced two-stream run and a run drained after every step.
Per step, copies of the action, the next field, the reward and the two flat gradient buffers are enqueued on the streams
that produce them (no host synchronisation in the un-synced run)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from det_probe import make, pkg  # noqa

def record(p, n, synced):
    red = pkg.distributed.GradReducer()
    gc, ga = red._view(p.policy.behavior_critic.model), red._view(p.policy.behavior_actor.model)
    hist = []
    for _ in range(n):
        k = p.tick
        p.run(1)
        with torch.cuda.stream(p.s_env):
            a, y, r = p.aring[k % 3].clone(), p.ybuf[(k + 1) % 2].clone(), p.rring[k % 3].clone()
        with torch.cuda.stream(p.s_upd):
            c, aa = gc.clone(), ga.clone()
        hist.append((a, y, r, c, aa))
        if synced:
            torch.cuda.synchronize()
    p.sync()
    return hist

n = int(os.environ.get("N", "150"))
ha = record(make(False), n, False)
hb = record(make(False), n, True)
names = ("action", "y_next", "reward", "critic_grad", "actor_grad")
found = False
for k, (x, y) in enumerate(zip(ha, hb)):
    bad = [(names[i], float((x[i] - y[i]).abs().max()), int((x[i] != y[i]).sum())) for i in range(5) if not torch.equal(x[i], y[i])]
    if bad:
        print(f"SPLIT={os.environ.get('PDEC_SPLIT')} KICK={os.environ.get('PDEC_KICK')} first difference at step {k}: {bad}")
        found = True
        # detail of the first differing tensor
        i = names.index(bad[0][0])
        d = (x[i] - y[i]).abs().flatten()
        idx = torch.nonzero(d).flatten()[:12].tolist()
        print("   differing flat indices (first 12):", idx, "of", d.numel(), " values:", [float(x[i].flatten()[j]) for j in idx[:4]], [float(y[i].flatten()[j]) for j in idx[:4]])
        break
if not found:
    print(f"SPLIT={os.environ.get('PDEC_SPLIT')} KICK={os.environ.get('PDEC_KICK')} no difference in {n} steps")
