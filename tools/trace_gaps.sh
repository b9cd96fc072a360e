#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/gaps_trace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps_trace -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline > $R/gpurun_out/gaps_trace.log 2>&1
F=$(ls $R/gpurun_out/gaps_trace/*/*kernel_trace.csv | head -1)
head -2 $F | cut -c1-400
python3 $R/tools/trace_gaps.py $F
rm -rf $R/gpurun_out/gaps_trace
