#!/bin/bash
# same-call A/B over environment switches: tools/ab_env.sh "<VAR=VAL> <VAR=VAL> ..." <command ...>; "-" = no variable; ABAB order
set -e
cd "$(dirname "$0")/.."
SPECS=$1; shift
for rep in 1 2; do
  for spec in - $SPECS; do
    echo "== $spec (pass $rep)"
    if [ "$spec" = "-" ]; then "$@"; else env "$spec" "$@"; fi
  done
done
