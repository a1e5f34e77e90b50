# config C5: number of batch parts (child environments) with the part streams on pipes of their own
run() { echo "== $*"; env "$@" python bench.py --config C5 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'], 2), round(d['ms_per_step'], 2))"; }
run PDEC_FLUID_SPLIT=2
run PDEC_FLUID_SPLIT=3
run PDEC_FLUID_SPLIT=4
run PDEC_FLUID_SPLIT=4 PDEC_BENCH_C5_PARTS=p,q,u
run PDEC_FLUID_SPLIT=3 PDEC_BENCH_C5_PARTS=p,q
run PDEC_FLUID_SPLIT=0
run PDEC_FLUID_SPLIT=2
