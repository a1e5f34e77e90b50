run() { echo "== $*"; env "$@" python bench.py --no-variants --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['repeat_ms_per_step'], d['issue']['calibration'])"; }
run PDEC_BENCH_STREAMS=lib
run PDEC_BENCH_STREAMS=lib
run PDEC_BENCH_STREAMS=torch
run PDEC_BENCH_STREAMS=torch
EXTRA="--issue eager" run PDEC_BENCH_STREAMS=lib
EXTRA="--issue eager" run PDEC_BENCH_STREAMS=torch
EXTRA="--repeats 4" run PDEC_BENCH_STREAMS=lib
EXTRA="--repeats 4" run PDEC_BENCH_ORDER=n,e,u,p,p
EXTRA="--repeats 4" run PDEC_BENCH_ORDER=n,e,u
EXTRA="--repeats 4" run PDEC_BENCH_ORDER=e,u
