#!/bin/bash
# the C2 control step with an emulated all-reduce latency of N us (one-workgroup spin kernel) where the
# collective of the actor gradient goes -- ON the update chain (--ar-order chain) and OFF it (--ar-order side: third stream,
# beside the next critic half).  One line per run: order, N, ms_per_step (median of the timed blocks), value.
# usage: bash tools/ar_chain_sweep.sh <out-file> [config-args...]
OUT=${1:-gpurun_out/ar_chain_sweep.txt}; shift
EXTRA="$@"
mkdir -p "$(dirname "$OUT")"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d.get('kernels_ms_per_step_in_pipeline',{})
print('$1', '$2', 'ms_per_step=%.5f' % d['ms_per_step'], 'value=%.0f' % d['value'], 'blocks=%s' % d['repeat_ms_per_step'], 'order=%s' % (d.get('allreduce_order') or 'fused single-GPU finish')[:24], 'finite=%s' % d['checks']['finite'])
"; }
echo "# bench.py $EXTRA --no-cpu-baseline --no-variants --steps 200 --repeats 5 [--emulate-ar-us N --ar-order O]" > "$OUT"
python bench.py $EXTRA --no-cpu-baseline --no-variants --steps 200 --repeats 5 2>/dev/null | line fused - >> "$OUT"
for N in 0 15 25 40 60; do
  for O in chain side; do
    python bench.py $EXTRA --no-cpu-baseline --steps 200 --repeats 5 --emulate-ar-us $N --ar-order $O 2>/dev/null | line $O $N >> "$OUT"
  done
done
python bench.py $EXTRA --no-cpu-baseline --no-variants --steps 200 --repeats 5 2>/dev/null | line fused - >> "$OUT"
cat "$OUT"
