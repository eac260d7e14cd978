#!/bin/bash
# round 4: phase timer of attention_pk_kernel<9> (tools build, hand-pipelined phases): where a pair's time goes, per wave
mkdir -p gpurun_out
{
for qb in 9 6 3; do
echo "== q_blocks $qb"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_PK_TIMER=1 CLIPX_ATTN_QBLOCKS=$qb timeout 120 tools/attn_bench 256 257 16 64 0
done
echo "== product"; timeout 120 tools/attn_bench 256 257 16 64 0
} > gpurun_out/r04t_attention_phases.log 2>&1
cat gpurun_out/r04t_attention_phases.log
