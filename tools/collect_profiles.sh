#!/bin/bash
# Collects the per-round profile artifacts on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02l      -> gpurun_out/<tag>_*; copy what is to be judged into profiles/
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench_default.json
python bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --issue eager > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_rocprof.log
cp $(ls $O/${TAG}_trace/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
python3 tools/trace_timeline.py $O/${TAG}_trace > $O/${TAG}_timeline.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --issue eager > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --issue eager > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_pmc_sq -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --issue eager > /dev/null 2>&1
python3 tools/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write > $O/${TAG}_pmc_traffic.json
python3 tools/pmc_sq_summary.py $O/${TAG}_pmc_sq > $O/${TAG}_sq_counters.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --two-layer 2>/dev/null | tail -1 > $O/${TAG}_bench_two_layer.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --replay 2>/dev/null | tail -1 > $O/${TAG}_bench_replay.json
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --nx 1024 2>/dev/null | tail -1 > $O/${TAG}_bench_c3_shard.json
PDEC_BENCH_BACKEND=gloo python bench.py --steps 100 --warmup 10 --no-cpu-baseline --gpus 2 2>/dev/null | tail -1 > $O/${TAG}_bench_n2_gloo_one_gpu.json
python tools/bench_rollout.py > $O/${TAG}_bench_rollout.jsonl 2>/dev/null
PDEC_SHARE=0 python bench.py --steps 600 --warmup 60 --no-cpu-baseline --issue eager 2>/dev/null | tail -1 > $O/${TAG}_ab_share0.json
python bench.py --steps 600 --warmup 60 --no-cpu-baseline --issue eager 2>/dev/null | tail -1 > $O/${TAG}_ab_share1.json
ls -la $O | grep ${TAG}_ | awk '{print $5, $9}'
