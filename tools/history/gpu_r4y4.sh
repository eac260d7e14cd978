#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/diag_i8.py 10000000 40000000 100000000 > gpurun_out/r04y4_diag_i8.log 2>&1
grep -v amdgpu.ids gpurun_out/r04y4_diag_i8.log | tail -40 | cut -c1-300
