#!/bin/bash
# runs tools/scan_time.py once per build under build_variants/ (and the default library); one JSON line each
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUTF=${1:-$ROOT/gpurun_out/variants.jsonl}
mkdir -p "$(dirname "$OUTF")"
shift
for lib in "$ROOT"/build_variants/*.so; do
    [ -e "$lib" ] || continue
    echo "== $lib" >&2
    HYPMERGE_LIB=$lib HM_VARIANT_TAG=$(basename "$lib" .so) timeout -k 10 240 python "$ROOT/tools/scan_time.py" "$@" >> "$OUTF" 2>> "$OUTF.err" || echo "{\"lib\": \"$lib\", \"failed\": true}" >> "$OUTF"
done
echo "variants done" >&2
