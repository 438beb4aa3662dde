#!/bin/bash
# unit-size / shape sweep of the bf16 scan's work loop (HM_TUNE_UNIT in 64-column units; 0 = static grid)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2d; mkdir -p $O; rm -f $O/sweep.jsonl
for shape in ${SHAPES:-0 1}; do
  for unit in ${UNITS:-0 4 8 16 32 64}; do
    HYPMERGE_LIB=$PWD/build_variants/shapes.so HM_VARIANT_TAG="s$shape u$unit" HM_TUNE_SHAPE=$shape HM_TUNE_UNIT=$unit timeout -k 10 120 python tools/scan_time.py --quick >> $O/sweep.jsonl 2>> $O/sweep.err || echo "{\"failed\": \"$shape $unit\"}" >> $O/sweep.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r2d/sweep.jsonl'):
    d=json.loads(l)
    if 'failed' in d: print(d); continue
    print(d['tag'], d.get('scan_ms_50000_bf16'), d.get('pflops_50000_bf16'), d.get('scan_ms_100000_bf16'), d.get('pflops_100000_bf16'), d.get('topk_scan_ms_50000_nocount'), d.get('topk_ms_50000_nocount'), d.get('pair_50000_bf16'))
PY
