#!/bin/bash
# round 4: the whole worker() pipeline with the reference-order reader (ViT-L/14), and one request stage by stage with the fused dedup
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/microbench.py pipeline > gpurun_out/r04d_pipeline_vitl14.log 2>&1; echo "pipeline rc=$?"; grep -v "^    stats" gpurun_out/r04d_pipeline_vitl14.log | tail -6
timeout 900 python tools/request_bench.py --reps 20 > gpurun_out/r04d_request.log 2>&1; echo "request rc=$?"
grep -v "^REQUEST" gpurun_out/r04d_request.log | tail -22
