#!/bin/bash
# Collects the per-round profile artifacts on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r03z [c2|c4|c5|b1|all]   -> gpurun_out/<tag>_*; copy what is to be judged into profiles/
set -u
TAG=${1:-rXX}
WHAT=${2:-all}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
SQ="SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
prof() {   # prof <name> <pmc-config> <bench args...>: kernel-trace + stats, then three counter-only passes
  local N=$1 CFG=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${N}trace -- python3 bench.py "$@" --no-cpu-baseline --no-variants > $O/${TAG}_${N}bench_under_rocprof.json 2> $O/${TAG}_${N}rocprof.log
  cp $(ls $O/${TAG}_${N}trace/*/*kernel_stats.csv | head -1) $O/${TAG}_${N}kernel_stats.csv
  [ "$N" = "" ] && python3 tools/trace_timeline.py $O/${TAG}_trace > $O/${TAG}_timeline.txt
  rm -rf $O/${TAG}_${N}trace
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_${N}pmc_fetch -- python3 bench.py "$@" --no-cpu-baseline --no-variants > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_${N}pmc_write -- python3 bench.py "$@" --no-cpu-baseline --no-variants > /dev/null 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d $O/${TAG}_${N}pmc_sq -- python3 bench.py "$@" --no-cpu-baseline --no-variants > /dev/null 2>&1
  PDEC_PMC_CONFIG=$CFG python3 tools/pmc_traffic.py $O/${TAG}_${N}pmc_fetch $O/${TAG}_${N}pmc_write > $O/${TAG}_${N}pmc_traffic.json
  PDEC_PMC_CONFIG=$CFG python3 tools/pmc_sq_summary.py $O/${TAG}_${N}pmc_sq > $O/${TAG}_${N}sq_counters.json
  rm -rf $O/${TAG}_${N}pmc_fetch $O/${TAG}_${N}pmc_write $O/${TAG}_${N}pmc_sq
}
if [ "$WHAT" = "all" ] || [ "$WHAT" = "c2" ]; then
  prof "" C2 --steps 100 --warmup 10 --issue eager
  J() { tail -1; }      # the JSON line is the last line of stdout (native libraries' chatter goes to stderr, bench.py protect_stdout)
  python bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-variants 2>/dev/null | J > $O/${TAG}_bench.json
  PDEC_SHARE=1 python bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-variants 2>/dev/null | J > $O/${TAG}_bench_step_in_64vgpr_form.json
  # timing independence of the results: exact f32 (the product) must give 0 differing run lengths
  ( N=120 SIDE_A="SYNC=1" SIDE_B="" python tools/det_probe5.py
    N=120 SIDE_A="PDEC_FINISH_REF=1" SIDE_B="" python tools/det_probe5.py
    N=60 B=512 E=51 SIDE_A="SYNC=1" SIDE_B="" python tools/det_probe5.py
  ) 2>&1 | grep "^\[" | cut -c1-400 > $O/${TAG}_timing_independence.txt
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --integrator rk4_fd 2>/dev/null | J > $O/${TAG}_bench_rk4_fd.json
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --two-layer 2>/dev/null | J > $O/${TAG}_bench_two_layer.json
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --replay 2>/dev/null | J > $O/${TAG}_bench_replay.json
  python bench.py --config C3 --no-cpu-baseline 2>/dev/null | J > $O/${TAG}_bench_c3_shard.json
  python bench.py --config C3 --no-cpu-baseline --overlap on 2>/dev/null | J > $O/${TAG}_bench_c3_shard_overlapped.json
  # what N > 1 adds to the update chain besides the wire: the data-parallel launch sequence on a 1-rank RCCL communicator
  python bench.py --steps 200 --warmup 20 --repeats 5 --split-update --no-cpu-baseline 2>/dev/null | J > $O/${TAG}_bench_split_update_policy.json
  python bench.py --steps 200 --warmup 20 --repeats 5 --split-update --dp-sync all --no-cpu-baseline 2>/dev/null | J > $O/${TAG}_bench_split_update_all.json
  PDEC_BENCH_BACKEND=gloo python bench.py --steps 100 --warmup 10 --no-cpu-baseline --gpus 2 2>/dev/null | J > $O/${TAG}_bench_n2_gloo_one_gpu.json
  python tools/bench_rollout.py > $O/${TAG}_bench_rollout.jsonl 2>/dev/null
  ./tools/fp64_issue_micro > $O/${TAG}_issue_micro.txt 2>/dev/null
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "c4" ]; then
  # the profiled C4 runs keep the whole batch in ONE launch per RK4 sub-step (PDEC_KSEG2D_SPLIT=0), so that the per-launch
  # figures -- kernel_stats' average duration, PMC traffic -- are those of the launch `roofline.achieved` is quoted on (168 MB
  # algorithmic); the timed bench lines below run the default three batch parts on three streams
  export PDEC_KSEG2D_SPLIT=0
  prof c4_ C4 --config C4 --steps 20 --warmup 12
  unset PDEC_KSEG2D_SPLIT
  python bench.py --config C4 2>/dev/null | tail -1 > $O/${TAG}_c4_bench.json
  python bench.py --config C4 --no-overlap --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_c4_bench_one_stream.json
  PDEC_BENCH_BACKEND=gloo python bench.py --config C4 --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_c4_bench_n2_gloo_one_gpu.json
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "c5" ]; then
  export PDEC_FLUID_SPLIT=0      # profiled runs: whole-batch launches, as for C4 above (the timed lines run two half-batch children)
  prof c5_ C5 --config C5 --steps 1 --warmup 1
  unset PDEC_FLUID_SPLIT
  python bench.py --config C5 2>/dev/null | tail -1 > $O/${TAG}_c5_bench.json
  PDEC_BENCH_BACKEND=gloo python bench.py --config C5 --gpus 2 --batch 4 --nx 128 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_c5_bench_n2_gloo_one_gpu_128.json
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "b1" ]; then
  # the reference-shaped single-trajectory training loop (KS22: B = 1, 20 x 3 updates per step) through run(): rate, then the kernel
  # statistics of the same command
  for w in ks22 kseg fluid; do python3 tools/b1_probe.py $w --two-streams 2>/dev/null | tail -3; done > $O/${TAG}_b1_probe.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_b1trace -- python3 tools/b1_probe.py ks22 --two-streams > /dev/null 2> $O/${TAG}_b1_rocprof.log
  cp $(ls $O/${TAG}_b1trace/*/*kernel_stats.csv | head -1) $O/${TAG}_b1_ks22_kernel_stats.csv
  rm -rf $O/${TAG}_b1trace
fi
# the driver's own command last (headline + variants + CPU baselines): the PMC traffic files this build's lines cite must be
# in profiles/ for roofline.traffic -- copy gpurun_out/${TAG}_*pmc_traffic.json there and re-run `python bench.py` if they were not
python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench_default.json
ls -la $O | grep ${TAG}_ | awk '{print $5, $9}'
