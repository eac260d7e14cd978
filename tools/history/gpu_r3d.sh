#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_preprocess_gpu.py tests/test_service_gpu.py -x -q -m gpu > gpurun_out/r3d_preprocess.log 2>&1; echo "preprocess rc=$?"
tail -5 gpurun_out/r3d_preprocess.log
timeout 600 python tools/resize_bench.py > gpurun_out/r3d_resize_bench.log 2>&1; echo "resize_bench rc=$?"
grep -v "^RESIZE" gpurun_out/r3d_resize_bench.log | tail -8
