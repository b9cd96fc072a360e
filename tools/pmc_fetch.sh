#!/bin/bash
# HBM-side bytes per launch (FETCH_SIZE / WRITE_SIZE passes only) of one model under one library build:
#   tools/pmc_fetch.sh <tag> <lib.so|cur> <model> <batch> <bits>     -> gpurun_out/<tag>_fetch.txt
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; LIBSRC=$2; MODEL=$3; BATCH=$4; BITS=$5
LIB=$R/diff-vit_amd/csrc/libp2vit_hip.so
cp $LIB /tmp/p2v_keep.so
trap 'cp /tmp/p2v_keep.so $LIB' EXIT
if [ "$LIBSRC" != cur ]; then cp $R/$LIBSRC $LIB; fi
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --model $MODEL --bits $BITS --batch $BATCH > $OUT/$c.log 2>&1 || tail -3 $OUT/$c.log
done
python3 $R/tools/pmc_summary.py $OUT > $R/gpurun_out/${TAG}_fetch.txt
rm -rf $OUT
grep -A3 "k_gemm_dma\|k_ln_gemm2\|k_lis" $R/gpurun_out/${TAG}_fetch.txt | head -60
