#!/bin/bash
# Build the engine library of a git revision (or of the working tree: rev = WORK) into a side file for same-call A/B runs (tools/abx.sh):
#   tools/build_rev.sh <rev|WORK> <out.so> [extra hipcc flags, e.g. -DP2V_NT_STORES=0]
# The product package never loads these files (tools/exp/ is git-ignored; the .so files travel to the GPU box with the snapshot).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=$1; OUT=$(realpath -m "$2"); shift 2
T=$(mktemp -d /tmp/p2v_rev.XXXXXX)
trap 'rm -rf $T' EXIT
mkdir -p $T/diff-vit_amd/csrc $T/include "$(dirname "$OUT")"
if [ "$REV" = WORK ]; then
  cp $ROOT/diff-vit_amd/csrc/*.hip $ROOT/diff-vit_amd/csrc/*.h $ROOT/diff-vit_amd/csrc/*.cpp $ROOT/diff-vit_amd/csrc/Makefile $T/diff-vit_amd/csrc/
  cp $ROOT/include/*.h $T/include/
else
  git -C $ROOT archive $REV diff-vit_amd/csrc include | tar -x -C $T
fi
make -s -C $T/diff-vit_amd/csrc -j8 LIB="$OUT" EXTRA="$*" 2>&1 | grep -E "error|warning" || true
ls -la "$OUT"
