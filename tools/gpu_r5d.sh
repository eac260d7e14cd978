#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
CLIPX_LIB=libclipx_ablate.so timeout 300 tools/gemm_bench -r 6 65536,3072,1024,0 65792,3072,1024,23 65792,4096,1024,17 65792,1024,1024,6 65792,1024,4096,6 -- 3:16 > $OUT/r05d_gemm_phases_direct.log 2>&1; cat $OUT/r05d_gemm_phases_direct.log
