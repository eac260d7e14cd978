#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
L=clip-retrieval_amd/lib
{ python tools/ab_knn_ring.py $L/libclipx_ablate.so; KNNX_RQ8_DBG=1 python tools/ab_knn_ring.py $L/libclipx_ablate.so; } > gpurun_out/r05k_rq8_nohits.log 2>&1
cat gpurun_out/r05k_rq8_nohits.log
