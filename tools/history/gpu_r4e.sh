#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
MB_PIPE_SHARDS=16 timeout 900 python tools/microbench.py pipeline > gpurun_out/r04g_pipeline_vitl14.log 2>&1; echo "pipeline rc=$?"; grep "pipeline" gpurun_out/r04g_pipeline_vitl14.log | tail -6
MB_PIPE_SHARDS=16 MB_PIPE_ONE_STREAM=1 timeout 900 python tools/microbench.py pipeline > gpurun_out/r04g_pipeline_vitl14_one_stream.log 2>&1; echo "pipeline rc=$?"; grep "pipeline" gpurun_out/r04g_pipeline_vitl14_one_stream.log | tail -6
