#!/bin/bash
# run one command under several library builds inside ONE gpurun call: tools/ab_cmd.sh "<label>=<lib.so> ..." <command ...>; "cur" = the in-tree library
set -e
cd "$(dirname "$0")/.."
LIB=diff-vit_amd/csrc/libp2vit_hip.so
cp $LIB /tmp/p2v_cur.so
trap 'cp /tmp/p2v_cur.so $LIB' EXIT
SPECS=$1; shift
for spec in cur=/tmp/p2v_cur.so $SPECS; do
  cp ${spec#*=} $LIB
  echo "== ${spec%%=*}"
  "$@"
done
