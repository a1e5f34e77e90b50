#!/bin/bash
set -u
python -m pytest tests/test_gpu_ks.py -m gpu -q -x -k "rk4_fd" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6
for r in 1 2; do python bench.py --steps 400 --warmup 40 --no-cpu-baseline --integrator rk4_fd 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); kp=d['kernels_ms_per_step_in_pipeline']; k=d['kernels_ms_per_step']; print('rk4_fd wave(dpp): ms/step %.4f' % d['ms_per_step'], '%.3f M' % (d['value']/1e6), {x[:10]:round(kp[x]*1e3,1) for x in kp}, 'alone step', round(k.get('ksfd_env_step',0)*1e3,1))"; done
