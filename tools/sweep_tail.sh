#!/bin/bash
# launch-geometry sweep of the static item grid: chunk size, share of the work cut into small tail items, tail divisor
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2h; mkdir -p $O; rm -f $O/sweep.jsonl
for shape in ${SHAPES:-0 1}; do
 for chunk in ${CHUNKS:-24 48 96}; do
  for tail in ${TAILS:-0.1 0.25 0.45}; do
   for div in ${DIVS:-2 4 8}; do
    HYPMERGE_LIB=$PWD/build_variants/shapes.so HM_VARIANT_TAG="s$shape c$chunk t$tail d$div" HM_TUNE_SHAPE=$shape HM_TUNE_CHUNK=$chunk HM_TUNE_TAIL=$tail HM_TUNE_TAIL_DIV=$div timeout -k 10 120 python tools/scan_time.py --quick >> $O/sweep.jsonl 2>> $O/sweep.err || echo "{\"failed\": \"$shape $chunk $tail $div\"}" >> $O/sweep.jsonl
   done
  done
 done
done
python - <<'PY'
import json
rows=[]
for l in open('gpurun_out/r2h/sweep.jsonl'):
    d=json.loads(l)
    if 'failed' in d: print(d); continue
    rows.append((d.get('scan_ms_50000_bf16'), d.get('scan_ms_100000_bf16'), d['tag']))
for r in sorted(rows): print(r)
PY
