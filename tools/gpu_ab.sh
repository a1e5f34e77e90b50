#!/bin/bash
set -u
export TAG=${1:-r03x}
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
tail -3 $O/${TAG}_pytest.log
for kick in 0 1; do PDEC_KICK=$kick N=600 python tools/det_probe4.py 2>&1 | grep -A1 "SPLIT="; done | tee $O/${TAG}_det4.txt
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --issue eager"
for rep in 1 2; do
  PDEC_SPLIT=0 $B 2>/dev/null | tail -1 > $O/${TAG}_ab_f32_$rep.json
  $B 2>/dev/null | tail -1 > $O/${TAG}_ab_split_$rep.json
  PDEC_KICK=0 $B 2>/dev/null | tail -1 > $O/${TAG}_ab_split_nokick_$rep.json
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s_ab_*.json" % os.environ["TAG"])):
    try:
        d=json.load(open(f)); k=d["kernels_ms_per_step"]; kp=d["kernels_ms_per_step_in_pipeline"]
        print(f.split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "alone:", {x:k[x] for x in ("ddpg_critic_fused","ddpg_actor_fused") if x in k}, "pipe:", {x:kp[x] for x in ("ddpg_critic_fused","ddpg_actor_fused","fused_finish","ks_env_step","policy_act_fused") if x in kp})
    except Exception as e: print(f, "ERR", e)
PY
