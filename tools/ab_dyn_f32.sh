# the item queue on the fp32 prefilter form (variant built with -DHM_DYN_F32=1) and on the bf16 top-k scan
set -e
export HYPMERGE_LIB=$PWD/build_variants/dynf32.so
echo "== argmin 50k d=100 fp32 form"; HM_SCAN_PRECISION=f32 AB_V=50000 timeout -k 10 250 python tools/ab_knobs.py static:dyn=0 dyn32:dyn=1 dyn48:dyn=1,chunk=48 dyn64:dyn=1,chunk=64 static48:dyn=0,chunk=48 2>&1 | grep -v amdgpu.ids
echo "== topk_count 50k bf16"; AB_V=50000 AB_MODE=topk_count timeout -k 10 250 python tools/ab_knobs.py static96:dyn=0,chunk=96 static128:dyn=0 dyn128:dyn=1 dyn96:dyn=1,chunk=96 2>&1 | grep -v amdgpu.ids
echo "== topk 50k bf16"; AB_V=50000 AB_MODE=topk timeout -k 10 250 python tools/ab_knobs.py static96:dyn=0,chunk=96 static128:dyn=0 dyn128:dyn=1 dyn96:dyn=1,chunk=96 2>&1 | grep -v amdgpu.ids
echo "== argmin 50k bf16 (final library defaults against the round's static grid)"; AB_V=50000 timeout -k 10 250 python tools/ab_knobs.py static96:dyn=0,chunk=96 queue128:dyn=1 2>&1 | grep -v amdgpu.ids
echo "== argmin 100k bf16"; AB_V=100000 timeout -k 10 250 python tools/ab_knobs.py static96:dyn=0,chunk=96 queue128:dyn=1 2>&1 | grep -v amdgpu.ids
