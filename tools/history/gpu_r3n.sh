#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/rq_sample_grid.py 0 256 > gpurun_out/r3n_rq_sample_grid.log 2>&1; echo "rc=$?"
grep "sample grid" gpurun_out/r3n_rq_sample_grid.log
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q -m gpu -k "rq or wide or coalesced or large_k or sharded" --durations=8 > gpurun_out/r3n_knn_tests.log 2>&1; echo "knn tests rc=$?"; tail -14 gpurun_out/r3n_knn_tests.log
