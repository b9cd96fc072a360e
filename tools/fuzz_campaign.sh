#!/bin/bash
# GPU box: a longer randomized-parity campaign than the one seed of tests/test_engine_gpu.py::test_fuzz_ops_against_oracle.
#   bash tools/fuzz_campaign.sh FIRST_SEED LAST_SEED CASES_PER_OP [LOG]
# 42 cases per op walk every (head_dim, token count) pair of the attention case once.  Stops at the first failing seed.
set -u
first=${1:-100}; last=${2:-111}; n=${3:-42}; log=${4:-gpurun_out/fuzz_campaign.txt}
mkdir -p "$(dirname "$log")"
: > "$log"
t0=$(date +%s)
for s in $(seq "$first" "$last"); do
  out=$(timeout -k 10 600 python tools/fuzz_ops.py "$s" "$n" 2>&1); rc=$?
  echo "seed $s rc=$rc $(echo "$out" | tail -1) [$(( $(date +%s) - t0 )) s]" | tee -a "$log"
  if [ $rc -ne 0 ]; then echo "$out" | tail -40 >> "$log"; exit 1; fi
done
echo "campaign: seeds $first..$last x $n cases per op: all equal" | tee -a "$log"
