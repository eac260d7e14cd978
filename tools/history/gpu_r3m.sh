#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/rq_sample_grid.py 256 128 64 32 16 > gpurun_out/r3m_rq_sample_grid.log 2>&1; echo "rc=$?"
grep "sample grid" gpurun_out/r3m_rq_sample_grid.log
