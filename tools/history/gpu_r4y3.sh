#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python bench.py --steps 2 --warmup 1 --no-parity --no-ab --cpu-seconds 0 --knn-batches 1,32,64,256 ) > gpurun_out/r04y3_bench_i8.log 2>&1
grep -v "^{" gpurun_out/r04y3_bench_i8.log | tail -30 | cut -c1-400
