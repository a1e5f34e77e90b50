# A/B of two builds of the library on config C4: bash tools/ab_lib.sh /path/to/other/libpdeconv.so
ALT=$1
run() { echo "== $*"; env "$@" python bench.py --config ${CFG:-C4} --no-variants --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'], 1), round(d['ms_per_step'], 4), r['kernel'], round(r['duration_us'], 2))"; }
for i in 1 2 3; do
run PDEC_LIB_PATH=$ALT
run X=1
done
