#!/bin/bash
# kernel trace of the GPU-bound (graph-replayed) default step: per-queue coverage + per-kernel average durations UNDER OVERLAP.
# usage (on the GPU box): bash tools/overlap_profile.sh <tag> [overlap_run.py arguments]
set -e
TAG=${1:-rXX}; shift || true
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_ovl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_ovl -- python3 $R/tools/overlap_run.py "$@" > $R/gpurun_out/${TAG}_ovl.log 2>&1 || tail -5 $R/gpurun_out/${TAG}_ovl.log
F=$(ls $R/gpurun_out/${TAG}_ovl/*/*kernel_trace.csv | head -1)
python3 $R/tools/trace_overlap.py $F > $R/gpurun_out/${TAG}_overlap.txt
grep -v amdgpu.ids $R/gpurun_out/${TAG}_ovl.log | tail -2 >> $R/gpurun_out/${TAG}_overlap.txt
rm -rf $R/gpurun_out/${TAG}_ovl
cat $R/gpurun_out/${TAG}_overlap.txt
