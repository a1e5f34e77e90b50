run() { echo "== $*"; env "$@" python bench.py --config C4 --no-variants --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'], 4))"; }
run PDEC_BENCH_PART_LEVEL=1
run PDEC_BENCH_PART_LEVEL=0
run PDEC_BENCH_PART_LEVEL=-1
run PDEC_BENCH_PART_LEVEL=-1 PDEC_BENCH_LEVELS=-1,-1
run PDEC_BENCH_PART_LEVEL=0 PDEC_BENCH_LEVELS=0,0
run PDEC_BENCH_PART_LEVEL=1 PDEC_BENCH_LEVELS=1,0
run PDEC_BENCH_PART_LEVEL=0 PDEC_BENCH_LEVELS=0,-1
run PDEC_BENCH_PART_LEVEL=1 PDEC_BENCH_LEVELS=1,-1
