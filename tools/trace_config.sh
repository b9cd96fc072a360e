#!/bin/bash
# rocprofv3 kernel-trace summary of one bench configuration: tools/trace_config.sh <tag> <bench.py arguments...>
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_trace.log 2>&1
cp $R/gpurun_out/${TAG}_trace/*/*kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $R/gpurun_out/${TAG}_trace
head -8 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-140
