#!/bin/bash
# round 4: attention at T = 257 -- 8 waves, one query block each, the 257th query split along the keys (product; tools build default)
# against the 6-wave kernel (tools build, CLIPX_ATTN_CFG=12); same box, alternating; then the attention / parity tests
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2 3; do
echo "== 8 waves (product)"; timeout 120 tools/attn_bench 256 257 16 64 0
echo "== 6 waves (tools build, cfg 12)"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=12 timeout 120 tools/attn_bench 256 257 16 64 0
done
echo "== B=1, B=32 (8 waves / 6 waves)"; timeout 120 tools/attn_bench 1 257 16 64 0; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=12 timeout 120 tools/attn_bench 1 257 16 64 0
timeout 120 tools/attn_bench 32 257 16 64 0; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=12 timeout 120 tools/attn_bench 32 257 16 64 0
echo "== T = 260 (6-wave kernel), T = 257 H = 12"; timeout 120 tools/attn_bench 64 260 16 64 0; timeout 120 tools/attn_bench 64 257 12 64 0
} > gpurun_out/r04x_attention_8wave.log 2>&1
cat gpurun_out/r04x_attention_8wave.log
timeout 900 python -m pytest tests/test_clip_gpu.py -m gpu -q -x -k "attention or parity_vs_oracle or full_depth or golden or ragged or pooled or large_batch or chunked" > gpurun_out/r04x_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04x_tests.log
