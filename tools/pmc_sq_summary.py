#!/usr/bin/env python3
"""Per-kernel means of SQ counters from a rocprofv3 --pmc run (separate pass from tracing, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) and the derived fractions:
  mfma_busy_over_sq_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES   (both as rocprofv3 reports them: summed over the
                           shader engines / XCDs that the counter aggregates; a ratio between builds, not an absolute)
  wait_any_frac    = SQ_WAIT_ANY / SQ_WAVE_CYCLES        (waves parked on s_waitcnt / barriers)
  wait_inst_frac   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (issue stalls)
  active_inst_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
Usage: python tools/pmc_sq_summary.py <rocprof output dir> > profiles/rNN_sq_counters.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

acc = defaultdict(lambda: defaultdict(list))
files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
for f in files:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pdec::", "").split("<")[0].strip()
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    e = {n: sum(v) / len(v) for n, v in c.items()}
    e["launches"] = max(len(v) for v in c.values())
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        for n, key in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_frac")):
            if n in e:
                e[key] = e[n] / wc
    if e.get("SQ_VALU_MFMA_BUSY_CYCLES") and e.get("SQ_BUSY_CYCLES"):
        e["mfma_busy_over_sq_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"]
    out[k] = e
json.dump({"csrc_sha16": bench.csrc_sha16(os.environ.get("PDEC_PMC_CONFIG", "C2")), "kernels": out}, sys.stdout, indent=1)
print()
