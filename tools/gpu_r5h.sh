#!/bin/bash
# round 5, visit h: counters of the int8 scan (4-wave and 8-wave forms) -- where the B = 256 pass spends its cycles
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
MB_KNN_ROWS=50000000 bash tools/pmc.sh r05h knn_rq8_scan knn > $ROOT/gpurun_out/r05h_rq8_pmc.txt 2>&1
cat $ROOT/gpurun_out/r05h_rq8_pmc.txt | cut -c1-1200
rm -rf $ROOT/gpurun_out/pmc_r05h/*/p_* 2>/dev/null
