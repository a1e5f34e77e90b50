# kernel trace of a few C4 steps with both kinds of pipeline streams: which HSA queue each launch went to, and the overlap
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in torch lib; do
  export PDEC_BENCH_STREAMS=$m
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/qtrace_$m -o t -- python3 $R/bench.py --config C4 --no-cpu-baseline --steps 4 --warmup 3 > $R/gpurun_out/qtrace_$m.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for m in ("torch", "lib"):
    f = glob.glob(f"gpurun_out/qtrace_{m}/**/*kernel_trace.csv", recursive=True)
    if not f: print(m, "no trace"); continue
    rows = list(csv.DictReader(open(f[0])))
    print(m, len(rows), "launches; columns:", list(rows[0].keys()))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tail = rows[-700:]
    t0 = int(tail[0]["Start_Timestamp"])
    q = collections.Counter((r["Queue_Id"], r.get("Stream_Id", "?"), r["Kernel_Name"][:28]) for r in tail)
    for k, v in sorted(q.items()): print("  ", k, v)
    for r in tail[-120:]:
        print("   %9.1f %9.1f q=%s s=%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"], r.get("Stream_Id", "?"), r["Kernel_Name"][:40]))
PY
