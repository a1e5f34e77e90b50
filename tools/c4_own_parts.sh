run() { echo "== $*"; env "$@" python bench.py --config C4 --no-variants --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'], 4))"; }
run PDEC_BENCH_ORDER=n,e,u
run PDEC_BENCH_ORDER=n,e,u PDEC_PART_LEVEL=1
run PDEC_BENCH_STREAMS=torch
run X=1
