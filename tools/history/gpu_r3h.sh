#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_service_gpu.py -x -q -m gpu -k "safety" > gpurun_out/r3h_safety.log 2>&1; echo "safety rc=$?"
tail -5 gpurun_out/r3h_safety.log
timeout 900 python tools/request_bench.py --reps 20 > gpurun_out/r3h_request.log 2>&1; echo "request rc=$?"
grep -v "^REQUEST" gpurun_out/r3h_request.log | tail -30
