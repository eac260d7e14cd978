#!/bin/bash
# round 5, visit c: direct-store epilogues (product lib) against the LDS-transposed ones (libclipx_ablate.so built with -DCLIPX_DIRECT_EPI=0)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SH="65792,3072,1024,23 65792,4096,1024,17 65792,1024,1024,6 65792,1024,4096,6 65536,3072,1024,0 19712,2304,768,23 65536,1024,1024,16"
{
for rnd in 1 2; do
echo "== direct (product)"; GEMM_BENCH_CHECKS=3 timeout 600 tools/gemm_bench -r 8 -b 30 $SH -- 3
echo "== lds-transposed (ablate lib, CLIPX_DIRECT_EPI=0)"; CLIPX_LIB=libclipx_ablate.so timeout 600 tools/gemm_bench -r 8 -b 30 $SH -- 3
done
} > $OUT/r05c_gemm_direct_epilogue.log 2>&1
grep -E "==|shape|cfg" $OUT/r05c_gemm_direct_epilogue.log
