#!/bin/bash
# round 5, visit f: kNN GPU tests with the tile-ordered / partial int8 copy, then the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1200 python -m pytest tests/test_knn_gpu.py -m gpu -x -q > $OUT/r05f_pytest_knn.log 2>&1; tail -15 $OUT/r05f_pytest_knn.log
( time timeout 900 python bench.py ) > $OUT/r05f_bench.log 2>&1; tail -4 $OUT/r05f_bench.log | cut -c1-3000
