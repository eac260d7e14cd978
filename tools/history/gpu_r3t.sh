#!/bin/bash
mkdir -p gpurun_out
{
for i in 1 2; do
echo "== 8 waves, last query key-split (product lib)"; timeout 120 tools/attn_bench 256 257 16 64 0
echo "== persistent 6 waves (cfg 12)"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=12 timeout 120 tools/attn_bench 256 257 16 64 0
echo "== one pair per workgroup (cfg 10)"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=10 timeout 120 tools/attn_bench 256 257 16 64 0
done
echo "== 8 waves B=1"; timeout 120 tools/attn_bench 1 257 16 64 0
echo "== 8 waves B=5 H=12"; timeout 120 tools/attn_bench 5 257 12 64 0
} > gpurun_out/r3t_attn_bench.log 2>&1
grep -v "^$" gpurun_out/r3t_attn_bench.log | tail -18
timeout 900 python -m pytest tests/test_clip_gpu.py -x -q -m gpu -k "attention or parity_vs_oracle or pooled or full_depth or large_batch or graphs" > gpurun_out/r3t_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3t_tests.log
