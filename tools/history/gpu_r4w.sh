#!/bin/bash
# round 4: phase timer of the final attention_pk_kernel<9> (tools build = product code + timer)
mkdir -p gpurun_out
{
echo "== final kernel, phase timer"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_PK_TIMER=1 timeout 120 tools/attn_bench 256 257 16 64 0
echo "== product"; timeout 120 tools/attn_bench 256 257 16 64 0
} > gpurun_out/r04w_attention_phases_final.log 2>&1
cat gpurun_out/r04w_attention_phases_final.log
