#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python -m pytest tests/test_knn_gpu.py -m gpu -x -q -k "i8 or rq or large_batch or register" > gpurun_out/r05j_pytest.log 2>&1; tail -4 gpurun_out/r05j_pytest.log
L=clip-retrieval_amd/lib
{ for r in 1 2; do python tools/ab_knn_ring.py $L/libclipx.so; python tools/ab_knn_ring.py $L/libclipx_ablate.so; done; } > gpurun_out/r05j_knn_loop_ab.log 2>&1
cat gpurun_out/r05j_knn_loop_ab.log
