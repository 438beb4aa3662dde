#!/bin/bash
# final_check.sh TAG: the round-end sequence on the GPU box -- the -m gpu suite, bench.py (defaults and the driver's
# flags), the rocprofv3 kernel trace of the bench, and the counter passes of the scan kernel -- outputs under gpurun_out/TAG_*
TAG=${1:-r03g}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp TQDM_DISABLE=1
set -o pipefail
echo "[1] item-queue tests"; timeout -k 10 300 python -m pytest tests/test_gpu_item_queue.py -x -q > $O/${TAG}_tests_queue.log 2>&1; rc=$?; tail -3 $O/${TAG}_tests_queue.log; [ $rc -eq 0 ] || exit 11
echo "[2] -m gpu suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_tests_gpu.log 2>&1; rc=$?; tail -3 $O/${TAG}_tests_gpu.log; [ $rc -eq 0 ] || exit 12
echo "[3] bench defaults"; timeout -k 10 400 python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err; rc=$?; tail -c 600 $O/${TAG}_bench_n1.json; [ $rc -eq 0 ] || exit 13
echo "[4] bench driver flags"; timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_n1_steps20_warmup5.json 2> $O/${TAG}_bench_s20.err; rc=$?; [ $rc -eq 0 ] || exit 14
echo "[5] rocprofv3 kernel trace"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_prof.err; rc=$?; [ $rc -eq 0 ] || exit 15
find $O/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
rm -rf $O/${TAG}_prof/*/*kernel_trace.csv
echo "[6] counter passes"; i=0; mkdir -p $O/${TAG}_pmc
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVES"; do
    i=$((i+1))
    timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${TAG}_pmc/pass$i -- python3 tools/scan_loop.py 5 > $O/${TAG}_pmc/pass$i.log 2>&1 \
        && echo "pass $i done: $grp" || { echo "pass $i FAILED: $grp"; break; }
done
python3 tools/pmc_summary.py $O/${TAG}_pmc_scan_kernel_bf16.json hm_scan_kernel $O/${TAG}_pmc/pass* > /dev/null 2>&1 && echo "pmc summary written"
echo "[done]"
