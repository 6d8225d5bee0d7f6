#!/bin/bash
# Build libfounddiff_hip.so of another git revision next to the tree's own (A/B runs: tools/probes/ab.sh).
#   bash tools/build_ref_lib.sh <git-rev> <out.so>        e.g.  bash tools/build_ref_lib.sh bfc1ae2 founddiff_amd/lib/ab/r4.so
set -eu
REV=$1; OUT=$2
T=$(mktemp -d)
git archive "$REV" founddiff_amd/csrc include | tar -x -C "$T"
mkdir -p "$(dirname "$OUT")" "$T/obj"
ls "$T"/founddiff_amd/csrc/*.hip | xargs -P 4 -I{} sh -c '/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -I'"$T"'/include -c {} -o '"$T"'/obj/$(basename {}).o'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$T"/obj/*.o
rm -rf "$T"
echo "$OUT"
