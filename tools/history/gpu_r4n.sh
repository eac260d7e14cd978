#!/bin/bash
# round 4: full GPU suite + the default bench line after the 16x16x32 switch
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r04n_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04n_pytest.log
( time timeout 900 python bench.py ) > gpurun_out/r04n_bench.log 2>&1; tail -4 gpurun_out/r04n_bench.log | cut -c1-1800
