#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_service.py tests/test_service_gpu.py -x -q -m gpu -k "range or dedup or large_k or request or knn_search" > gpurun_out/r3o_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3o_tests.log
timeout 900 python tools/request_bench.py --reps 20 > gpurun_out/r3o_request.log 2>&1; echo "request rc=$?"
grep -v "^REQUEST" gpurun_out/r3o_request.log | tail -19
