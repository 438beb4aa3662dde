# interleaved A/B of the scan's item queue (knobs dyn, dyn_line) and item-list geometries against the static grid;
# library: a quick variant from tools/build_variant.sh (d = 100 only).  Lists here: ONE round (or two) of long items that
# covers share s0 of the work, everything else in short items (ph_div1 / ph_div2).
set -e
export HYPMERGE_LIB=$PWD/build_variants/dynq.so
D="dyn=1,dyn_line=1,phases=3"
mk() {  # chunk s0 div1 div2
  s1=$(python3 -c "print(round((1-$2)*0.7,4))")
  echo "c$1_s$2_d$3_$4:$D,chunk=$1,ph_share0=$2,ph_share1=$s1,ph_div1=$3,ph_div2=$4"
}
SP50="static: dynA112:dyn=1,dyn_line=1,chunk=112"
for c in 112 128 144; do for s in 0.62 0.68 0.74 0.80; do SP50="$SP50 $(mk $c $s 8 16)"; done; done
for c in 128 144; do for s in 0.68 0.74; do SP50="$SP50 $(mk $c $s 6 12) $(mk $c $s 12 12) $(mk $c $s 16 16)"; done; done
SP50="$SP50 $(mk 160 0.80 10 20) $(mk 160 0.86 10 20) $(mk 176 0.86 11 22)"
echo "== argmin 50k"; AB_V=50000 timeout -k 10 250 python tools/ab_knobs.py $SP50 2>&1 | grep -v amdgpu.ids
SP100="static: dynA128:dyn=1,dyn_line=1,chunk=128"
for c in 112 128 144; do for s in 0.62 0.70 0.78; do SP100="$SP100 $(mk $c $s 4 8)"; done; done
for c in 224 256 288; do for s in 0.62 0.70 0.78; do SP100="$SP100 $(mk $c $s 8 16)"; done; done
SP100="$SP100 $(mk 128 0.70 8 16) $(mk 256 0.70 16 32) $(mk 256 0.78 16 32)"
echo "== argmin 100k"; AB_V=100000 timeout -k 10 250 python tools/ab_knobs.py $SP100 2>&1 | grep -v amdgpu.ids
