#!/bin/bash
set -u
TAG=${1:-r03ca}
O=gpurun_out
mkdir -p $O
python -m pytest tests/test_gpu_agent.py tests/test_gpu_mlp.py tests/test_gpu_pipeline.py -m gpu -q -x > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
grep -E "passed|failed|error|rc=" $O/${TAG}_pytest.log | tail -5
python tools/bench_rollout.py 2>/dev/null | tee $O/${TAG}_bench_rollout.jsonl | cut -c1-330
PDEC_ROLLOUT_PERSISTENT=0 python tools/bench_rollout.py 2>/dev/null | tee $O/${TAG}_bench_rollout_host_enqueued.jsonl | cut -c1-330
python bench.py --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
