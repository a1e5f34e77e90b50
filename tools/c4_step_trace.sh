# kernel trace of a few timed C4 steps: start / end of every launch of one control step, per hardware queue
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c4trace -o t -- python3 $R/bench.py --config C4 --no-cpu-baseline --no-variants --steps 6 --warmup 6 > $R/gpurun_out/c4trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c4trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "actuate" in r["Kernel_Name"]]
i0, i1 = idx[8], idx[9]
t0 = int(rows[i0]["Start_Timestamp"])
print("step", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "us")
for r in rows[i0 - 8:i1 + 1]:
    if "rk4" in r["Kernel_Name"]: continue
    print("  %8.1f %8.1f dur %6.1f q=%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Queue_Id"], r["Kernel_Name"][:60]))
rk = [r for r in rows[i0:i1] if "rk4" in r["Kernel_Name"]]
print("  rk4 launches", len(rk), "first start", (int(rk[0]["Start_Timestamp"]) - t0) / 1e3, "last end", max(int(r["End_Timestamp"]) for r in rk) / 1e3 - t0 / 1e3)
PY
