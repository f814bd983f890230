#!/bin/bash
# Build libarcq_hip.so of another git revision beside the working tree's (same-box A-B runs: tools/e2e_lib_ab.py).
# usage: tools/scripts/build_ref_lib.sh <git-ref> <out.so>      (run from the repository root, in the build container)
set -e
ref=$1; out=$(realpath -m $2); tmp=$(mktemp -d)
mkdir -p $tmp/a/b/c $tmp/include
git archive $ref arcquant_amd/csrc include | tar -x -C $tmp
# keep the relative include paths of the sources valid: csrc at depth 2 below the include directory's parent
make -C $tmp/arcquant_amd/csrc -j4 OUT=$out > $tmp/build.log 2>&1 || { tail -20 $tmp/build.log; exit 1; }
rm -rf $tmp
echo built $out from $ref
