#!/bin/bash
# round 4, final tree: rocprofv3 kernel-trace stats of the default bench command (no CPU legs)
mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04final -o r04final -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --no-parity --no-ab > $GRAFT_REPO_ROOT/gpurun_out/r04final_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/prof_r04final -name "*kernel_stats.csv" | head -1) gpurun_out/r04final_kernel_stats.csv
find gpurun_out/prof_r04final -name "*kernel_trace.csv" -delete
head -14 gpurun_out/r04final_kernel_stats.csv | cut -c1-150
