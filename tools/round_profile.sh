#!/bin/bash
# end-of-milestone evidence run on the GPU box: parity suite, default bench line, kernel-trace stats, PMC passes.
# usage: bash tools/round_profile.sh <tag>      (writes gpurun_out/<tag>_*;  SKIP_TESTS=1 skips the parity suite, OTHER_MODELS=1 adds the
#        bench lines + HBM-traffic passes of vit_base b512, deit_base W4 b256 and swin_base b256)
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
# the same launches ALONE: one 68-image slice per step on one stream (the launch size of the default run's slices) - the average durations of this
# trace are what bench.py's isolated `avg_launch_us` must agree with; in the trace above the slices overlap (worker threads keep four queues fed
# even under the profiler), so its averages are the under-overlap durations and its minima the isolated ones
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace1 -- python3 $R/bench.py --batch 68 --streams 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${TAG}_trace1.log 2>&1
cp $R/gpurun_out/${TAG}_trace1/*/*kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats_isolated.csv
head -8 $R/gpurun_out/${TAG}_kernel_stats_isolated.csv | cut -c1-150
rm -rf $R/gpurun_out/${TAG}_trace1
PMC_BATCH=68 bash $R/tools/pmc.sh > $R/gpurun_out/${TAG}_pmc.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc > $R/gpurun_out/${TAG}_pmc_summary.txt
cp $R/gpurun_out/pmc/summary.json $R/gpurun_out/${TAG}_pmc_summary.json
rm -rf $R/gpurun_out/pmc
grep -A20 "k_ln_gemm2<5, 6, 1, false>" $R/gpurun_out/${TAG}_pmc_summary.txt | head -24
if [ -n "$OTHER_MODELS" ]; then
  cd $R
  python bench.py --model vit_base --batch 512 --streams 2 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/${TAG}_cfg3_vit_base_b512.json 2>/dev/null
  python bench.py --model deit_base --bits 4 --batch 256 --streams 2 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/${TAG}_cfg5_deit_base_w4.json 2>/dev/null
  python bench.py --model swin_base --batch 256 --streams 3 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/${TAG}_cfg4_swin_base.json 2>/dev/null
  for f in cfg3_vit_base_b512 cfg5_deit_base_w4 cfg4_swin_base; do python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])" gpurun_out/${TAG}_$f.json; done
  cd /tmp
  for spec in "vit_base 256 8" "deit_base 128 4"; do
    set -- $spec
    OUT=$R/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --model $1 --bits $3 --batch $2 > $OUT/$c.log 2>&1 || tail -3 $OUT/$c.log
    done
    python3 $R/tools/pmc_summary.py $OUT > $R/gpurun_out/${TAG}_pmc_summary_$1.txt
    cp $OUT/summary.json $R/gpurun_out/${TAG}_pmc_summary_$1.json
    rm -rf $OUT
  done
fi
