#!/bin/bash
# round 4: the GEMMs on v_mfma_f32_16x16x32 (product library) against v_mfma_f32_32x32x16 (libclipx_ablate.so built with -DCLIPX_MFMA16=0)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_clip_gpu.py -m gpu -q -x -k "gemm or parity_vs_oracle or large_batch or chunked or pooled or ragged or full_depth_vit_l14 or golden" > gpurun_out/r04k_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r04k_tests.log
SHAPES="65792,3072,1024,23 65792,4096,1024,17 65792,1024,1024,6 65792,1024,4096,6 19712,2304,768,23 10547,768,3072,6"
{
echo "== product: 16x16x32"; timeout 300 tools/gemm_bench -r 10 $SHAPES -- 3
echo "== ablate lib: 32x32x16"; CLIPX_LIB=libclipx_ablate.so timeout 300 tools/gemm_bench -r 10 $SHAPES -- 3
echo "== product: 16x16x32"; timeout 300 tools/gemm_bench -r 10 $SHAPES -- 3
echo "== ablate lib: 32x32x16"; CLIPX_LIB=libclipx_ablate.so timeout 300 tools/gemm_bench -r 10 $SHAPES -- 3
} > gpurun_out/r04k_gemm_bench.log 2>&1
grep -E "^==|shape|median|bitwise|differ" gpurun_out/r04k_gemm_bench.log | cut -c1-120
