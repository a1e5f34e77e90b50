# config C5 (fluid, call-by-call loop on the null stream, second half of the batch on a part stream): where the part stream's queue sits
run() { echo "== $*"; env "$@" AMD_LOG_LEVEL=4 AMD_LOG_MASK=16 python bench.py --config C5 --no-cpu-baseline --steps 3 --warmup 1 2>/tmp/q.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'], 2), round(d['ms_per_step'], 2))"; grep "Created SWq" /tmp/q.txt | sed 's/.*with priority \([0-9]\).*/\1/' | tr '\n' ' '; echo; }
run PDEC_BENCH_STREAMS=torch
run PDEC_BENCH_ORDER=
run PDEC_BENCH_ORDER=p
run PDEC_BENCH_ORDER=d0,p
run PDEC_BENCH_ORDER=d0,d0,p
run PDEC_BENCH_ORDER=d0,d0,d0,p
run PDEC_BENCH_ORDER=p PDEC_BENCH_PART_LEVEL=0
run PDEC_BENCH_ORDER=d0,d0,p PDEC_BENCH_PART_LEVEL=0
run PDEC_BENCH_STREAMS=torch PDEC_FLUID_SPLIT=0
