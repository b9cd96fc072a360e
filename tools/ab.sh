#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (box-to-box variance is larger than most kernel changes):
#   tools/ab.sh <baseline.so> [bench.py arguments]   -> alternates baseline / current, prints images/s of every run
set -e
cd "$(dirname "$0")/.."
LIB=diff-vit_amd/csrc/libp2vit_hip.so
cp $LIB /tmp/p2v_new.so
trap 'cp /tmp/p2v_new.so $LIB' EXIT      # a failing run must not leave the baseline binary installed
BASE=$1; shift
for r in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then cp $BASE $LIB; else cp /tmp/p2v_new.so $LIB; fi
    python bench.py --no-cpu-baseline --repeats 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
  done
done
cp /tmp/p2v_new.so $LIB
