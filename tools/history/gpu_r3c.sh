#!/bin/bash
# round 3, third GPU call: the resize/crop kernel (row f2) -- parity tests, microbench, then the whole GPU suite
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_preprocess_gpu.py -x -q -m gpu > gpurun_out/r3c_preprocess.log 2>&1; echo "preprocess rc=$?"
tail -15 gpurun_out/r3c_preprocess.log
timeout 600 python tools/resize_bench.py > gpurun_out/r3c_resize_bench.log 2>&1; echo "resize_bench rc=$?"
tail -12 gpurun_out/r3c_resize_bench.log
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3c_pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r3c_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r3c_bench.log 2>&1; echo "bench rc=$?"
tail -3 gpurun_out/r3c_bench.log
