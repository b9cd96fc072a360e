#!/bin/bash
# A/B/C... of library builds and environment switches inside ONE gpurun call (box-to-box variance is larger than most kernel changes).
#   tools/abx.sh "<label>=<lib.so>[,VAR=VAL...] ..." [bench.py arguments]      "cur" as the library = the in-tree build
# Two alternating rounds; prints label, images/s and ms per step of every run.
set -e
cd "$(dirname "$0")/.."
LIB=diff-vit_amd/csrc/libp2vit_hip.so
cp $LIB /tmp/p2v_cur.so
trap 'cp /tmp/p2v_cur.so $LIB' EXIT
SPECS=$1; shift
for r in 1 2; do
  for spec in $SPECS; do
    label=${spec%%=*}; rest=${spec#*=}
    lib=${rest%%,*}; envs=""
    if [ "$rest" != "$lib" ]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
    if [ "$lib" = cur ]; then lib=/tmp/p2v_cur.so; fi
    cp $lib $LIB
    env $envs python bench.py --no-cpu-baseline --repeats 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$label', d['value'], d['ms_per_step'])"
  done
done
