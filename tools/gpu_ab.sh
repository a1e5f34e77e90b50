#!/bin/bash
# A/B session for the update-chain kernels: tests of the touched paths, phase stamps, back-to-back bench pairs
set -u
export TAG=${1:-r03b}
O=gpurun_out
mkdir -p $O
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_mlp.py tests/test_gpu_agent.py -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
tail -4 $O/${TAG}_pytest.log
for pf in 0 1; do
  PDEC_PREFETCH=$pf PDEC_STAMPS=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --issue eager 2>&1 >/dev/null | grep "pdec stamps" | tail -3 > $O/${TAG}_stamps_pf$pf.txt
  tail -1 $O/${TAG}_stamps_pf$pf.txt
done
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --issue eager"
for rep in 1 2; do
  PDEC_PREFETCH=0 PDEC_FINISH_REF=1 $B 2>/dev/null | tail -1 > $O/${TAG}_ab_pf0_ref_$rep.json
  PDEC_PREFETCH=0 $B 2>/dev/null | tail -1 > $O/${TAG}_ab_pf0_new_$rep.json
  PDEC_PREFETCH=1 PDEC_FINISH_REF=1 $B 2>/dev/null | tail -1 > $O/${TAG}_ab_pf1_ref_$rep.json
  $B 2>/dev/null | tail -1 > $O/${TAG}_ab_pf1_new_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/%s_ab_*.json" % __import__("os").environ["TAG"])):
    try:
        d=json.load(open(f)); k=d["kernels_ms_per_step"]; kp=d["kernels_ms_per_step_in_pipeline"]
        print(f.split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "alone:", {x:k[x] for x in ("ddpg_critic_fused","ddpg_actor_fused","fused_finish") if x in k}, "pipe:", {x:kp[x] for x in ("ddpg_critic_fused","ddpg_actor_fused","fused_finish","ks_env_step") if x in kp})
    except Exception as e: print(f, "ERR", e)
PY
