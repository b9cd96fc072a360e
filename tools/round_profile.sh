#!/bin/bash
# end-of-milestone evidence run on the GPU box: parity suite, default bench line, kernel-trace stats, PMC passes.
# usage: bash tools/round_profile.sh <tag>      (writes gpurun_out/<tag>_*)
set -e
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd $R
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests -x -q -m gpu 2>&1 | tail -2; fi
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 2000 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_trace.log 2>&1
cp $R/gpurun_out/${TAG}_trace/*/*kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
head -12 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
rm -rf $R/gpurun_out/${TAG}_trace      # the raw trace is tens of MB; only the summary is kept
bash $R/tools/pmc.sh > $R/gpurun_out/${TAG}_pmc.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc > $R/gpurun_out/${TAG}_pmc_summary.txt
cp $R/gpurun_out/pmc/summary.json $R/gpurun_out/${TAG}_pmc_summary.json
rm -rf $R/gpurun_out/pmc
grep -A16 "k_ln_gemm<5, 6>" $R/gpurun_out/${TAG}_pmc_summary.txt | head -40
