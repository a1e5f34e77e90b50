#!/usr/bin/env python3
"""Generate tests/golden/nn_torch_golden.npz: an independent torch-autograd (CPU, fp64)
evaluation of the DDPG losses and gradients of src/PDEagent.jl:385-409 on seeded inputs.
The NN path has no reference artifact that pins it, so this cross-check and
finite differences pin the oracle.  Run: python tests/golden/make_nn_golden.py"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)
rng = np.random.default_rng(7)
ns, na, h, H, Bu = 3, 1, 16, 140, 37
dA, dC = [ns, h, h, na], [ns + na, H, H, 1]


def mk(d):
    P = []
    for i in range(3):
        lim = np.sqrt(6.0 / (d[i] + d[i + 1]))
        P += [rng.uniform(-lim, lim, (d[i + 1], d[i])), rng.standard_normal(d[i + 1]) * 0.1]
    return P


def fwd(P, x, last):
    h1 = torch.relu(P[0] @ x + P[1][:, None])
    h2 = torch.relu(P[2] @ h1 + P[3][:, None])
    z = P[4] @ h2 + P[5][:, None]
    return torch.tanh(z) if last == "tanh" else z


out = {}
nets = {k: mk(d) for k, d in (("A", dA), ("C", dC), ("At", dA), ("Ct", dC))}
for k, P in nets.items():
    for i, p in enumerate(P):
        out[f"{k}{i}"] = p
s, sn = rng.standard_normal((ns, Bu)), rng.standard_normal((ns, Bu))
a, r = rng.uniform(-1, 1, (na, Bu)), -rng.uniform(0, 1, Bu)
t = (rng.uniform(0, 1, Bu) < 0.2) * 1.0
out.update(s=s, sn=sn, a=a, r=r, t=t)
T = {k: [torch.tensor(p, requires_grad=True) for p in P] for k, P in nets.items()}
ts, tsn, ta, tr, tt = map(torch.tensor, (s, sn, a, r, t))
with torch.no_grad():
    qt = fwd(T["Ct"], torch.cat([tsn, fwd(T["At"], tsn, "tanh")]), None).reshape(-1)
for quirk in (1, 0):
    q = fwd(T["C"], torch.cat([ts, ta]), None).reshape(-1)
    if quirk:   # (1 x Bu) .+ (Bu)  ->  Bu x Bu, src/PDEagent.jl:388,393
        qnext = tr[None, :] + (0.99 * (1 - tt) * qt)[:, None]
        loss = ((qnext - q[:, None]) ** 2).mean()
    else:
        loss = ((tr + 0.99 * (1 - tt) * qt - q) ** 2).mean()
    gs = torch.autograd.grad(loss, T["C"])
    out[f"closs_q{quirk}"] = loss.item()
    for i, gi in enumerate(gs):
        out[f"gC{i}_q{quirk}"] = gi.numpy()
aloss = -fwd(T["C"], torch.cat([ts, fwd(T["A"], ts, "tanh")]), None).mean()
gs = torch.autograd.grad(aloss, T["A"])
out["aloss"] = aloss.item()
for i, gi in enumerate(gs):
    out[f"gA{i}"] = gi.numpy()
np.savez_compressed(os.path.join(HERE, "nn_torch_golden.npz"), **out)
print("wrote nn_torch_golden.npz", len(out), "arrays")
