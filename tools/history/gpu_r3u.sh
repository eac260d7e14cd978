#!/bin/bash
mkdir -p gpurun_out
{
echo "== one pair per workgroup (cfg 10)"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=10 timeout 120 tools/attn_bench 256 257 16 64 0
echo "== persistent 6 waves (cfg 12)"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=12 timeout 120 tools/attn_bench 256 257 16 64 0
echo "== 8 waves small"; timeout 120 tools/attn_bench 1 257 16 64 0
echo "== one pair per workgroup again (cfg 10)"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=10 timeout 120 tools/attn_bench 256 257 16 64 0
echo "== 8 waves (product lib)"; timeout 120 tools/attn_bench 256 257 16 64 0
} > gpurun_out/r3u_attn_bench.log 2>&1
grep -v "^$" gpurun_out/r3u_attn_bench.log | tail -14
