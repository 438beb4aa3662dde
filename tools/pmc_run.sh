#!/bin/bash
# Counter passes for the pair-scan kernel (one rocprofv3 run per counter group; --kernel-trace only;
# FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage (on the GPU box): bash tools/pmc_run.sh OUTDIR [env assignments for the target, e.g. HM_SCAN_PRECISION=f32]
OUT=$1; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 tools/scan_loop.py 5 > "$OUT/pass$i.log" 2>&1 \
        && echo "pass $i done: $grp" || echo "pass $i FAILED: $grp"
done
