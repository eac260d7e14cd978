#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_clip_gpu.py tests/test_service_gpu.py -x -q -m gpu -k "pooled or large_batch or parity_vs_oracle or safety or graphs or attention" > gpurun_out/r3i_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r3i_tests.log
timeout 300 python tools/ab_encode.py CLIPX_FULL_LAST_BLOCK 1 0 3 > gpurun_out/r3i_ab_pool.log 2>&1; tail -3 gpurun_out/r3i_ab_pool.log
MB_PIPE_MODEL=ViT-B/32 MB_READER_SAMPLES=4000 timeout 900 python tools/microbench.py reader pipeline > gpurun_out/r3i_reader_pipeline_b32.log 2>&1; echo "mb rc=$?"
grep -v "^    stats" gpurun_out/r3i_reader_pipeline_b32.log | tail -24
