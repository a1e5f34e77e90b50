# config C4 (or CFG=C5 / C2) under different creation orders of its streams (bench.py, pipeline_streams): value, ms per step, and
# the priority (2 high, 1 normal, 0 low) of the hardware queues in the order the HIP runtime made them.  Tokens: e env stream,
# u update stream, p part stream, n first use of the null stream, d<level> a stream that stays idle.
run() { echo "== $*"; env "$@" AMD_LOG_LEVEL=4 AMD_LOG_MASK=16 python bench.py --config ${CFG:-C4} --no-variants --no-cpu-baseline 2>/tmp/q.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'], 4))"; grep "Created SWq" /tmp/q.txt | sed 's/.*with priority \([0-9]\).*/\1/' | tr '\n' ' '; echo; }
run PDEC_BENCH_STREAMS=torch
run PDEC_BENCH_ORDER=
run PDEC_BENCH_ORDER=n,e,u,p,p
run PDEC_BENCH_ORDER=e,n,u,p,p
run PDEC_BENCH_ORDER=e,u,n,p,p
run PDEC_BENCH_ORDER=d0,e,u,p,p
run PDEC_BENCH_ORDER=d0,d0,e,u,p,p
run PDEC_BENCH_ORDER=d0,d0,d0,e,u,p,p
run PDEC_BENCH_ORDER=d-1,d1,e,u,p,p
run PDEC_BENCH_ORDER=e,u,p,d0,p
run PDEC_BENCH_ORDER=e,d0,d0,d0,u,p,p
run PDEC_BENCH_ORDER=e,u,p
run PDEC_BENCH_ORDER=e,u
