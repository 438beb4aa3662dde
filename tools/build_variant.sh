#!/bin/bash
# build_variant.sh NAME "-DFOO=1 -DBAR=2": a variant of libhypmerge.so (d = 100 instantiations only) under
# build_variants/NAME.so, for side-by-side timing on the GPU box (tools/scan_variants.sh)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build_variants"
make -s -C "$ROOT/hyptokenizer_amd/csrc" -j4 OBJDIR="$ROOT/build_variants/.obj_$NAME" OUT="$ROOT/build_variants/$NAME.so" \
     EXTRA="-DHM_SCAN_INSTANTIATE_ALL=0 -DHM_TUNING=1 $*"
echo "built build_variants/$NAME.so"
