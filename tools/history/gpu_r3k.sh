#!/bin/bash
mkdir -p gpurun_out
{
for i in 1 2; do
echo "== asm-placed LDS reads (product lib)"; timeout 120 tools/attn_bench 256 257 16 64 0
echo "== asm-placed, ablate lib default"; CLIPX_LIB=libclipx_ablate.so timeout 120 tools/attn_bench 256 257 16 64 0
echo "== C++ reads (cfg 10, ablate lib)"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=10 timeout 120 tools/attn_bench 256 257 16 64 0
done
echo "== text T=77 causal"; timeout 120 tools/attn_bench 256 77 12 64 1
echo "== B/32 image T=50"; timeout 120 tools/attn_bench 256 50 12 64 0
echo "== B/16 image T=197"; timeout 120 tools/attn_bench 256 197 12 64 0
echo "== H/14 dh 80"; timeout 120 tools/attn_bench 64 257 16 80 0
} > gpurun_out/r3k_attn_bench.log 2>&1
cat gpurun_out/r3k_attn_bench.log | grep -v "^$" | tail -40
timeout 900 python -m pytest tests/test_clip_gpu.py -x -q -m gpu -k "attention or parity_vs_oracle or pooled or full_depth or large_batch" > gpurun_out/r3k_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3k_tests.log
