#!/bin/bash
# round 4: attention_pk_kernel<9> with the hand-pipelined S phase (product) against hipcc's schedule (libclipx_ablate.so built with
# -DCLIPX_ATTN_SPIPE=0): same box, alternating; then the attention tests
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for rep in 1 2 3; do
echo "== S phase pipelined by hand"; timeout 120 tools/attn_bench 256 257 16 64 0
echo "== hipcc's schedule"; CLIPX_LIB=libclipx_ablate.so timeout 120 tools/attn_bench 256 257 16 64 0
done
echo "== B=1 (pipelined / hipcc)"; timeout 120 tools/attn_bench 1 257 16 64 0; CLIPX_LIB=libclipx_ablate.so timeout 120 tools/attn_bench 1 257 16 64 0
} > gpurun_out/r04s_attention_spipe.log 2>&1
cat gpurun_out/r04s_attention_spipe.log
timeout 900 python -m pytest tests/test_clip_gpu.py -m gpu -q -x -k "attention or parity_vs_oracle or full_depth or golden or ragged" > gpurun_out/r04s_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04s_tests.log
