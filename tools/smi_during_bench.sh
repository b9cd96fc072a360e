#!/bin/bash
# samples clocks / power with rocm-smi (1 Hz) while a long default bench loop runs (is the step power- or clock-limited?)
# usage (GPU box): bash tools/smi_during_bench.sh <tag> [bench.py arguments]
TAG=${1:-rXX}; shift || true
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --no-cpu-baseline --steps 2000 --repeats 5 "$@" > gpurun_out/${TAG}_smi_bench.json 2>/dev/null &
BP=$!
: > gpurun_out/${TAG}_smi.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power \(W\)|GPU use" | sed -e 's/.*: //' | tr '\n' ' ' >> gpurun_out/${TAG}_smi.txt
  echo >> gpurun_out/${TAG}_smi.txt
  sleep 1
done
wait $BP
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_smi_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
sort -t' ' -k1 gpurun_out/${TAG}_smi.txt | uniq -c | sort -rn | head -12
