#!/bin/bash
# kernel trace (rocprofv3) of the C2 control step with an emulated all-reduce of $1 us in the order $2 (chain | side):
# start offset / duration / hardware queue of every launch of three steady-state steps (tools/trace_timeline.py)
N=${1:-25}; ORD=${2:-side}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/artrace_$ORD
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/artrace_$ORD -o t -- python3 $R/bench.py --no-cpu-baseline --steps 120 --warmup 20 --emulate-ar-us $N --ar-order $ORD > $R/gpurun_out/artrace_$ORD.log 2>&1
cd $R
python tools/trace_timeline.py gpurun_out/artrace_$ORD > gpurun_out/artrace_${ORD}_timeline.txt 2>&1
head -60 gpurun_out/artrace_${ORD}_timeline.txt
rm -rf gpurun_out/artrace_$ORD
