#!/bin/bash
# bench lines of the three configs with the PMC traffic files of this build in place (roofline.traffic non-null)
set -u
O=gpurun_out; mkdir -p $O; export TAG=${1:-r04}
python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench_default.json
python bench.py --config C4 2>/dev/null | tail -1 > $O/${TAG}_c4_bench.json
python bench.py --config C5 2>/dev/null | tail -1 > $O/${TAG}_c5_bench.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --integrator rk4_fd 2>/dev/null | tail -1 > $O/${TAG}_bench_rk4_fd.json
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s_*.json" % os.environ["TAG"])):
    d=json.load(open(f))
    if "roofline" not in d: continue
    r=d["roofline"]
    print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],4), r["kernel"], round(r["frac"],3), "traffic", r.get("traffic"))
PY
