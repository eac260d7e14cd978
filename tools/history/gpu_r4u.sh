#!/bin/bash
# round 4: attention_pk_kernel<9>, new dealing of blocks / DMAs to the waves (product) against the old one (libclipx_ablate.so built
# with -DCLIPX_ATTN_ROLES=0 (r04u) or the state of commit 2 (r04v)), both with the hand-pipelined phases; same box, alternating; phase timer of the old dealing; tests
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2 3; do
echo "== new roles"; timeout 120 tools/attn_bench 256 257 16 64 0
echo "== old roles"; CLIPX_LIB=libclipx_ablate.so timeout 120 tools/attn_bench 256 257 16 64 0
done
echo "== B=1, B=32 (new / old)"; timeout 120 tools/attn_bench 1 257 16 64 0; CLIPX_LIB=libclipx_ablate.so timeout 120 tools/attn_bench 1 257 16 64 0
timeout 120 tools/attn_bench 32 257 16 64 0; CLIPX_LIB=libclipx_ablate.so timeout 120 tools/attn_bench 32 257 16 64 0
} > gpurun_out/r04v_attention_roles.log 2>&1
cat gpurun_out/r04v_attention_roles.log
timeout 900 python -m pytest tests/test_clip_gpu.py -m gpu -q -x -k "attention or parity_vs_oracle or full_depth or golden or ragged or pooled" > gpurun_out/r04v_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04v_tests.log
