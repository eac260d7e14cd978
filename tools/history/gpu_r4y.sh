#!/bin/bash
# round 4: int8 first stage of the flat scans -- its tests, then the kNN legs of the bench with and without it
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_knn_gpu.py -m gpu -q -x -k "rq or i8" > gpurun_out/r04y_i8_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r04y_i8_tests.log | cut -c1-300
