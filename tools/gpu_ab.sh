#!/bin/bash
# session for the update-chain kernels: tests of the touched paths, phase stamps, bench runs, kernel trace
set -u
export TAG=${1:-r03c}
O=gpurun_out
mkdir -p $O
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_mlp.py tests/test_gpu_agent.py tests/test_aa_multirank_gpu.py -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
tail -4 $O/${TAG}_pytest.log
PDEC_STAMPS=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --issue eager 2>&1 >/dev/null | grep "pdec stamps" | tail -3 > $O/${TAG}_stamps.txt
tail -1 $O/${TAG}_stamps.txt
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --issue eager"
for rep in 1 2 3; do
  $B 2>/dev/null | tail -1 > $O/${TAG}_ab_new_$rep.json
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --issue eager > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_rocprof.log
cp $(ls $O/${TAG}_trace/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
python3 tools/trace_timeline.py $O/${TAG}_trace > $O/${TAG}_timeline.txt
rm -rf $O/${TAG}_trace
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s_ab_*.json" % os.environ["TAG"])):
    try:
        d=json.load(open(f)); k=d["kernels_ms_per_step"]; kp=d["kernels_ms_per_step_in_pipeline"]
        print(f.split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "alone:", {x:k[x] for x in ("ddpg_critic_fused","ddpg_actor_fused","fused_finish") if x in k}, "pipe:", {x:kp[x] for x in ("ddpg_critic_fused","ddpg_actor_fused","fused_finish","ks_env_step") if x in kp})
    except Exception as e: print(f, "ERR", e)
PY
head -12 $O/${TAG}_kernel_stats.csv | cut -c1-150
tail -25 $O/${TAG}_timeline.txt
