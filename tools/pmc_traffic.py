#!/usr/bin/env python3
"""Turn rocprofv3 --pmc output into per-launch HBM traffic of each kernel.

Collect (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"
prescribe -- FETCH_SIZE and WRITE_SIZE do not fit one pass, and never together with tracing):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_traffic.json

Units and gfx950 corrections applied (same guide): the counters are in KiB; FETCH_SIZE tallies the
128-B fabric requests of a wide coalesced stream at 64 B, i.e. reports HALF the bytes -> `fetch_bytes_x2`
is the corrected figure for 16-B-per-lane streaming reads and `fetch_bytes_raw` the uncorrected one
(narrower accesses are uncalibrated: the truth lies between the two); WRITE_SIZE is exact for
16-B-per-lane stores.  bench.py reads the newest profiles/r*_pmc_traffic.json for `roofline.traffic`.
For `bench.py --config C4 | C5` collect with that flag, set PDEC_PMC_CONFIG=C4 | C5 for this script (the kernel-source hash it
stamps is the one of that config) and name the file profiles/rNN_c4_pmc_traffic.json / rNN_c5_pmc_traffic.json.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(dirname):
    acc = defaultdict(lambda: defaultdict(list))
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel-Name") or ""
                cn, cv = row.get("Counter_Name"), row.get("Counter_Value")
                if cn is None or cv is None:
                    continue
                acc[name][cn].append(float(cv))
    return acc, files


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("pdec::", "")
    return n.split("<")[0].strip()


def main():
    out = {}
    srcs = []
    for d in sys.argv[1:]:
        acc, files = collect(d)
        srcs += files
        for name, ctrs in acc.items():
            e = out.setdefault(short(name), {"full_name": name})
            for cn, vals in ctrs.items():
                e[cn + "_KiB_mean"] = sum(vals) / len(vals)
                e[cn + "_launches"] = len(vals)
    for e in out.values():
        f, w = e.get("FETCH_SIZE_KiB_mean"), e.get("WRITE_SIZE_KiB_mean")
        if f is not None:
            e["fetch_bytes_raw"] = f * 1024
            e["fetch_bytes_x2"] = 2 * f * 1024
        if w is not None:
            e["write_bytes"] = w * 1024
        if f is not None and w is not None:
            e["hbm_bytes_per_launch"] = 2 * f * 1024 + w * 1024       # corrected as the guide prescribes
            e["hbm_bytes_per_launch_uncorrected"] = f * 1024 + w * 1024
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                                      # csrc_sha16(): which kernel sources these counters belong to
    json.dump({"source_files": [os.path.relpath(s) for s in srcs], "csrc_sha16": bench.csrc_sha16(os.environ.get("PDEC_PMC_CONFIG", "C2")), "kernels": out},
              sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
