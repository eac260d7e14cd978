#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_clip_gpu.py -x -q -m gpu -k "pooled or large_batch or parity_vs_oracle or graphs or full_depth" > gpurun_out/r3e_pool.log 2>&1; echo "pool tests rc=$?"
tail -6 gpurun_out/r3e_pool.log
timeout 600 python tools/ab_encode.py CLIPX_FULL_LAST_BLOCK 1 0 4 > gpurun_out/r3e_ab_pool.log 2>&1; echo "ab rc=$?"
tail -4 gpurun_out/r3e_ab_pool.log
