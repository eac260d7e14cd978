#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_knn_gpu.py -m gpu -q -x -k "rq or i8" > gpurun_out/r04y7_i8_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r04y7_i8_tests.log | cut -c1-300
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04y7 -o p -- python $GRAFT_REPO_ROOT/tools/diag_i8.py 100000000 > $GRAFT_REPO_ROOT/gpurun_out/r04y7_diag.log 2>&1
cd $GRAFT_REPO_ROOT; grep -v amdgpu.ids gpurun_out/r04y7_diag.log | tail -6 | cut -c1-200
head -8 $(find gpurun_out/prof_r04y7 -name "*kernel_stats.csv" | head -1) | cut -c1-200
find gpurun_out/prof_r04y7 -name "*kernel_trace.csv" -delete
