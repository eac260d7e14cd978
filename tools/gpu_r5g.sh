#!/bin/bash
# round 5, visit g: full GPU test suite, then the default bench line with the new legs (125 M shard, anisotropic corpus, host paths, ivf)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r05g_pytest_gpu.log 2>&1; tail -5 $OUT/r05g_pytest_gpu.log
( time timeout 900 python bench.py ) > $OUT/r05g_bench.log 2>&1; tail -6 $OUT/r05g_bench.log | cut -c1-1500
