#!/bin/bash
# shape / chunk sweep of the bf16 scan
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b; mkdir -p $O; rm -f $O/sweep.jsonl
for lib in shapes shapes_sub2; do
 for shape in ${SHAPES:-0 1 3}; do
  for chunk in 48 96 192; do
    HYPMERGE_LIB=$PWD/build_variants/$lib.so HM_VARIANT_TAG="$lib s$shape c$chunk" HM_TUNE_SHAPE=$shape HM_TUNE_CHUNK=$chunk timeout -k 10 120 python tools/scan_time.py --quick >> $O/sweep.jsonl 2>> $O/sweep.err || echo "{\"failed\": \"$lib $shape $chunk\"}" >> $O/sweep.jsonl
  done
 done
done
python - <<'PY'
import json
for l in open('gpurun_out/r2b/sweep.jsonl'):
    d=json.loads(l)
    if 'failed' in d: print(d); continue
    print(d['tag'], d.get('scan_ms_50000_bf16'), d.get('pflops_50000_bf16'), d.get('scan_ms_100000_bf16'), d.get('pflops_100000_bf16'), d.get('topk_scan_ms_50000_nocount'), d.get('topk_ms_50000_nocount'))
PY
