#!/bin/bash
# A/B/C of library builds inside ONE gpurun call: tools/ab3.sh "<label>=<lib.so> ..." [bench.py arguments]; "cur" = the in-tree library
set -e
cd "$(dirname "$0")/.."
LIB=diff-vit_amd/csrc/libp2vit_hip.so
cp $LIB /tmp/p2v_cur.so
trap 'cp /tmp/p2v_cur.so $LIB' EXIT
SPECS=$1; shift
for r in 1 2; do
  for spec in cur=/tmp/p2v_cur.so $SPECS; do
    cp ${spec#*=} $LIB
    python bench.py --no-cpu-baseline --repeats 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('${spec%%=*}', d['value'], d['ms_per_step'])"
  done
done
