#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
L=clip-retrieval_amd/lib
{ for r in 1 2; do python tools/ab_knn_ring.py $L/libclipx.so; python tools/ab_knn_ring.py $L/libclipx_ablate.so; done; } > gpurun_out/r05i_knn_ring8_vs_ring4.log 2>&1
cat gpurun_out/r05i_knn_ring8_vs_ring4.log
