#!/bin/bash
# round 5, visit b: where the encoder's GEMMs lose against the plain bf16 GEMM of the same shape (operand type, epilogue, ragged tail),
# isolated + sustained, and the phase timer of the encoder's own forms
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
G="timeout 600 tools/gemm_bench -r 12 -b 30"
{
echo "== QKV"; $G 65536,3072,1024,0 65536,3072,1024,16 65536,3072,1024,23 65792,3072,1024,23 -- 3
echo "== fc1"; $G 65536,4096,1024,0 65536,4096,1024,1 65536,4096,1024,16 65536,4096,1024,17 65792,4096,1024,17 -- 3
echo "== out_proj"; $G 65536,1024,1024,0 65536,1024,1024,6 65792,1024,1024,6 -- 3
echo "== fc2"; $G 65536,1024,4096,0 65536,1024,4096,6 65792,1024,4096,6 -- 3
} > $OUT/r05b_gemm_decomposition.log 2>&1
cat $OUT/r05b_gemm_decomposition.log
CLIPX_LIB=libclipx_ablate.so timeout 300 tools/gemm_bench -r 6 65536,3072,1024,0 65792,3072,1024,23 65792,4096,1024,17 65792,1024,1024,6 65792,1024,4096,6 -- 3:16 > $OUT/r05b_gemm_phases.log 2>&1; cat $OUT/r05b_gemm_phases.log
