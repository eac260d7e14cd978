#!/bin/bash
# round 5, visit a: same-box / same-data A/B of the library GEMM vs the vendor's (isolated + sustained), the clocks and MFMA-busy
# counters of both, their L2-miss traffic, and the phase timer of the four encoder GEMMs
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 600 python tools/ab_vendor.py > $OUT/r05a_ab_vendor.log 2>&1; cat $OUT/r05a_ab_vendor.log
cd /tmp
AB_REPS=3 AB_BURST=10 timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/r05a_pmc1 -o p -- python $ROOT/tools/ab_vendor.py > $OUT/r05a_pmc1.log 2>&1
python3 $ROOT/tools/pmc_summary.py $(find $OUT/r05a_pmc1 -name 'p_counter_collection.csv' | head -1) gemm Cijk > $OUT/r05a_ab_vendor_clock_mfma.txt 2>&1; cat $OUT/r05a_ab_vendor_clock_mfma.txt
AB_REPS=3 AB_BURST=4 timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/r05a_pmc2 -o p -- python $ROOT/tools/ab_vendor.py > $OUT/r05a_pmc2.log 2>&1
python3 $ROOT/tools/pmc_summary.py $(find $OUT/r05a_pmc2 -name 'p_counter_collection.csv' | head -1) gemm Cijk > $OUT/r05a_ab_vendor_fetch.txt 2>&1; cat $OUT/r05a_ab_vendor_fetch.txt
cd $ROOT
CLIPX_LIB=libclipx_ablate.so timeout 300 tools/gemm_bench -r 10 65792,3072,1024,23 65792,4096,1024,17 65792,1024,1024,6 65792,1024,4096,6 -- 3 3:16 > $OUT/r05a_gemm_phases.log 2>&1; cat $OUT/r05a_gemm_phases.log
rm -rf $OUT/r05a_pmc1 $OUT/r05a_pmc2
