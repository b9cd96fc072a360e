#!/bin/bash
# same-call comparison of batch slicings of the default bench: tools/ab_slices.sh "<streams>:<slices>[:extra bench flags] ..."   (two alternating rounds)
cd "$(dirname "$0")/.."
for r in 1 2; do
  for spec in "$@"; do
    st=${spec%%:*}; rest=${spec#*:}; sl=${rest%%:*}; extra=""
    if [ "$rest" != "$sl" ]; then extra=$(echo "${rest#*:}" | tr '+' ' '); fi
    python bench.py --no-cpu-baseline --repeats 3 --steps 30 --streams $st --slices $sl $extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('streams $st slices $sl $extra', d['value'], d['ms_per_step'])"
  done
done
