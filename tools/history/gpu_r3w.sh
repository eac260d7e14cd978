#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_clip_gpu.py tests/test_service_gpu.py -x -q -m gpu -k "ragged or parity_vs_oracle or pooled or full_depth or large_batch or graphs or chunked or async or pipelined or mapper or worker or checkpoint" > gpurun_out/r3w_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3w_tests.log
timeout 300 python tools/ab_encode.py CLIPX_RAGGED_TEXT 0 1 3 > gpurun_out/r3w_ab_ragged.log 2>&1; tail -3 gpurun_out/r3w_ab_ragged.log
