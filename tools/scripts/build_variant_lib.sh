#!/bin/bash
# Build libarcq_hip.so of the WORKING TREE with extra compiler flags into another file (timing experiments: A-B against the product
# library with tools/tile_lib_ab.py / tools/e2e_lib_ab.py).  usage: tools/scripts/build_variant_lib.sh "<flags>" <out.so>
set -e
flags=$1; out=$(realpath -m $2); tmp=$(mktemp -d)
mkdir -p $tmp/arcquant_amd $tmp/include
cp -r arcquant_amd/csrc $tmp/arcquant_amd/csrc
cp include/*.h $tmp/include/
rm -rf $tmp/arcquant_amd/csrc/_build
make -C $tmp/arcquant_amd/csrc -j8 OUT=$out EXTRA="$flags" > $tmp/build.log 2>&1 || { tail -20 $tmp/build.log; exit 1; }
rm -rf $tmp
echo built $out with "$flags"
