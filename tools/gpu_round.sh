#!/bin/bash
# one GPU-box session: tests, then the bench lines of every config (outputs under gpurun_out/<tag>_*)
set -u
TAG=${1:-r03a}
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
tail -5 $O/${TAG}_pytest.log
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -c 600 $O/${TAG}_bench_default.json
python bench.py --integrator rk4_fd --no-cpu-baseline > $O/${TAG}_bench_rk4_fd.json 2> $O/${TAG}_bench_rk4_fd.err; tail -c 300 $O/${TAG}_bench_rk4_fd.err
python bench.py --config C4 > $O/${TAG}_bench_c4.json 2> $O/${TAG}_bench_c4.err; tail -c 300 $O/${TAG}_bench_c4.err
python bench.py --config C5 > $O/${TAG}_bench_c5.json 2> $O/${TAG}_bench_c5.err; tail -c 300 $O/${TAG}_bench_c5.err
ls -la $O | grep ${TAG}_ | awk '{print $5, $9}'
