#!/bin/bash
# round 4: store cache-policy ablation on the bf16-output epilogue (QKV / fc1 shaped GEMMs), ablation library
mkdir -p gpurun_out
export CLIPX_LIB=libclipx_ablate.so
timeout 600 tools/gemm_bench -r 10 65792,3072,1024,0 65792,4096,1024,0 19712,2304,768,0 -- 3 3:31 3:32 3:33 3 3:31 > gpurun_out/r04h_gemm_store_policy.log 2>&1
cat gpurun_out/r04h_gemm_store_policy.log | tail -40
