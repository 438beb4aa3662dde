#!/bin/bash
# persistent equal-share grid (HM_TUNE_PERSIST = 1: occupancy query, 1 + N: N blocks per CU) vs static item grid (0)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2g; mkdir -p $O; rm -f $O/sweep.jsonl
for shape in ${SHAPES:-0 1}; do
  for persist in ${PERSIST:-0 1 2 3 4}; do
    HYPMERGE_LIB=$PWD/build_variants/shapes.so HM_TUNE_VERBOSE=1 HM_VARIANT_TAG="s$shape p$persist" HM_TUNE_SHAPE=$shape HM_TUNE_PERSIST=$persist timeout -k 10 120 python tools/scan_time.py --quick >> $O/sweep.jsonl 2>> $O/sweep.err || echo "{\"failed\": \"$shape $persist\"}" >> $O/sweep.jsonl
  done
done
grep hypmerge $O/sweep.err | sort | uniq -c
python - <<'PY'
import json
for l in open('gpurun_out/r2g/sweep.jsonl'):
    d=json.loads(l)
    if 'failed' in d: print(d); continue
    print(d['tag'], d.get('scan_ms_50000_bf16'), d.get('pflops_50000_bf16'), d.get('scan_ms_100000_bf16'), d.get('pflops_100000_bf16'), d.get('topk_scan_ms_50000_nocount'), d.get('topk_ms_50000_nocount'))
PY
