#!/bin/bash
set -u
export TAG=${1:-r03z}
O=gpurun_out
mkdir -p $O
python -m pytest tests/test_gpu_fluid.py tests/test_gpu_mlp.py tests/test_gpu_pipeline.py -m gpu -q -x > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
tail -3 $O/${TAG}_pytest.log
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --issue eager"
for rep in 1 2; do $B 2>/dev/null | tail -1 > $O/${TAG}_ab_new_$rep.json; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s_ab_*.json" % os.environ["TAG"])):
    d=json.load(open(f)); k=d["kernels_ms_per_step"]; kp=d["kernels_ms_per_step_in_pipeline"]; r=d["roofline"]
    print(f.split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "alone:", {x:k[x] for x in ("ddpg_critic_fused","ddpg_actor_fused","fused_finish") if x in k}, "pipe:", {x:kp[x] for x in ("ddpg_critic_fused","ddpg_actor_fused","fused_finish","ks_env_step") if x in kp}, "busy %.3f clk %.3f"%(r["mfma_busy_frac"], r["shader_clock_GHz"]), r["phase_cycles"])
PY
for w in fluid kseg ks22; do PDEC_FLUID_GRAPH=1 python tools/b1_probe.py $w 2>/dev/null | head -1; done
PDEC_FLUID_GRAPH=0 python tools/b1_probe.py fluid 2>/dev/null | head -1 | sed 's/^/graph off: /'
