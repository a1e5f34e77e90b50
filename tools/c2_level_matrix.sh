# config C2: queue priority levels of the env stream and the update stream (PDEC_BENCH_LEVELS=env,update)
run() { echo "== $*"; env "$@" python bench.py --no-variants --no-cpu-baseline --repeats 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['repeat_ms_per_step'], d['roofline'].get('duration_us_in_pipeline'))"; }
run PDEC_BENCH_LEVELS=-1,0
run PDEC_BENCH_LEVELS=0,-1
run PDEC_BENCH_LEVELS=0,0
run PDEC_BENCH_LEVELS=-1,-1
run PDEC_BENCH_LEVELS=1,-1
run PDEC_BENCH_LEVELS=1,0
run PDEC_BENCH_LEVELS=-1,0
