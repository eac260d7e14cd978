#!/bin/bash
# round 4: attention_pk_kernel<9>, time of a launch against the query blocks computed per pair (tools build, CLIPX_ATTN_QBLOCKS)
mkdir -p gpurun_out
{
for rep in 1 2; do
for qb in 9 6 3 1 8 7; do echo "== q_blocks $qb"; CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_QBLOCKS=$qb timeout 120 tools/attn_bench 256 257 16 64 0 | sed 's/max |err.*//'; done
done
} > gpurun_out/r04r_attention_qblocks.log 2>&1
cat gpurun_out/r04r_attention_qblocks.log
