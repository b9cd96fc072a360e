#!/bin/bash
# rocprofv3 PMC passes over a short single-stream bench run at the launch size of the default bench (one slice of the default run, PMC_BATCH images:
# the same kernels and shapes as the sliced run, without a concurrent kernel polluting the counters) (counters in their own runs: no trace domains besides kernel-trace)
# usage (GPU box): [PMC_MODEL=deit_small] [PMC_BATCH=86] [PMC_BITS=8] bash tools/pmc.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
rm -rf $OUT
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --model ${PMC_MODEL:-deit_small} --bits ${PMC_BITS:-8} --batch ${PMC_BATCH:-86} > $OUT/$name.log 2>&1 || tail -5 $OUT/$name.log; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run sq2 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
ls $OUT/*/* | head -30
