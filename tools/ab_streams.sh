#!/bin/bash
# same-call ABAB of --streams values for one bench configuration: tools/ab_streams.sh "<bench flags>" <streams> <streams> ...
cd "$(dirname "$0")/.."
FLAGS=$1; shift
for r in 1 2; do
  for st in "$@"; do
    python bench.py $FLAGS --no-cpu-baseline --repeats 3 --steps 10 --warmup 3 --streams $st 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$FLAGS streams $st', d['config'].get('batch_slices'), d['value'], d['ms_per_step'])"
  done
done
