#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/diag_i8.py 100000000 > gpurun_out/r04y5_diag_i8.log 2>&1
grep -v amdgpu.ids gpurun_out/r04y5_diag_i8.log | tail -12 | cut -c1-200
bash tools/gpu_r4y2.sh
