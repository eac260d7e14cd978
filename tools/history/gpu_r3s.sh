#!/bin/bash
mkdir -p gpurun_out
{
for i in 1 2; do
echo "== persistent 6 waves (ablate lib default)"; CLIPX_LIB=libclipx_ablate.so timeout 120 tools/attn_bench 256 257 16 64 0
echo "== persistent 5 waves (cfg 11)"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=11 timeout 120 tools/attn_bench 256 257 16 64 0
echo "== one pair per workgroup (cfg 10)"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=10 timeout 120 tools/attn_bench 256 257 16 64 0
done
} > gpurun_out/r3s_attn_bench.log 2>&1
grep -v "^$" gpurun_out/r3s_attn_bench.log | tail -14
