#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3g -o r3g -- python $R/bench.py --steps 1 --warmup 1 --knn-batches 256 --knn-scans 5 --cpu-seconds 0 --no-parity > $R/gpurun_out/r3g_rocprof.log 2>&1; echo "rc=$?"
cd $R
tail -2 gpurun_out/r3g_rocprof.log | cut -c1-600
f=$(find gpurun_out/prof_r3g -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-200
cp $f gpurun_out/r3g_kernel_stats.csv
