#!/bin/bash
mkdir -p gpurun_out
{
for i in 1 2; do
echo "== persistent (product lib)"; timeout 120 tools/attn_bench 256 257 16 64 0
echo "== one head per workgroup (cfg 10, ablate lib)"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=10 timeout 120 tools/attn_bench 256 257 16 64 0
done
echo "== persistent B=1"; timeout 120 tools/attn_bench 1 257 16 64 0
echo "== persistent B=3 T=250"; timeout 120 tools/attn_bench 3 250 16 64 0
} > gpurun_out/r3p_attn_bench.log 2>&1
cat gpurun_out/r3p_attn_bench.log | grep -v "^$" | tail -16
timeout 900 python -m pytest tests/test_clip_gpu.py -x -q -m gpu -k "attention or parity_vs_oracle or pooled or full_depth or large_batch or graphs" > gpurun_out/r3p_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r3p_tests.log
