#!/bin/bash
# round 5, visit e: stagger (odd workgroups take their tail strip first) A/B inside the tools build: flags 2 = off, 10 = on
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
SH="65792,3072,1024,23 65792,4096,1024,17 65792,1024,1024,6 65792,1024,4096,6"
{
CLIPX_LIB=libclipx_ablate.so GEMM_BENCH_CHECKS=2 timeout 600 tools/gemm_bench -r 10 -b 30 $SH -- 3:0:2 3:0:10 3:16:2 3:16:10
} > $OUT/r05e_gemm_stagger.log 2>&1
cat $OUT/r05e_gemm_stagger.log
