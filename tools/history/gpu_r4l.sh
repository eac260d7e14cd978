#!/bin/bash
mkdir -p gpurun_out
{
echo "### 3072 unit scale"; DIAG_UNIT=1 timeout 300 python tools/diag_gemm_variants.py 65792 3072 1024 7
echo "### 1024"; timeout 300 python tools/diag_gemm_variants.py 65792 1024 1024 7
echo "### 4096"; timeout 300 python tools/diag_gemm_variants.py 65792 4096 1024 7
echo "### 3072 K=768"; timeout 300 python tools/diag_gemm_variants.py 65792 3072 768 7
} > gpurun_out/r04l_diag.log 2>&1
grep -v amdgpu.ids gpurun_out/r04l_diag.log
