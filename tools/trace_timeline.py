#!/usr/bin/env python3
"""Diagnostic: steady-state timeline of one control step from a rocprofv3 --kernel-trace CSV (kernel start / end per
stream): per kernel the start offset inside the step, its duration, and the gaps on the update chain."""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pdec::", "")[:40],
                         r.get("Queue_Id", "?")))
rows.sort()
crit = [i for i, r in enumerate(rows) if r[2].startswith("ddpg_critic_fused") or r[2].startswith("ddpg2_critic")]
if len(crit) < 30:
    print("too few steps", len(crit)); sys.exit(0)
# steady-state control steps = the longest run of consecutive critic passes that each have an actor pass and a PDE step
# before the next one (the bench's kernel-timing passes replay single kernels back to back and are skipped this way)
def full_step(i):
    names = [r[2] for r in rows[crit[i]:crit[i + 1]]]
    return any(n.startswith(("ddpg_actor_fused", "ddpg2_actor")) for n in names) and any("env_step" in n for n in names)
best, cur = (0, 0), None
for i in range(len(crit) - 1):
    if full_step(i):
        cur = (cur[0], i + 1) if cur else (i, i + 1)
        if cur[1] - cur[0] > best[1] - best[0]:
            best = cur
    else:
        cur = None
mid = (best[0] + best[1]) // 2
sel = crit[mid: mid + 8]
print("step period (critic start -> next critic start), us:", [round((rows[sel[i + 1]][0] - rows[sel[i]][0]) / 1e3, 1) for i in range(len(sel) - 1)])
for a, b in zip(sel[:3], sel[1:4]):
    t0 = rows[a][0]
    print("---- step")
    for r in rows[a:b]:
        print(f"  +{(r[0] - t0) / 1e3:7.1f} us  dur {(r[1] - r[0]) / 1e3:6.1f}  q{r[3]:>3}  {r[2]}")
per = defaultdict(list)
for r in rows[crit[best[0]]:crit[max(best[0] + 1, best[1] - 1)]]:
    per[r[2]].append((r[1] - r[0]) / 1e3)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:42s} n={len(v):5d} mean {sum(v) / len(v):7.1f} us")
