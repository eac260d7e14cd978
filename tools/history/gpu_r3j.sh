#!/bin/bash
mkdir -p gpurun_out
MB_PIPE_MODEL=ViT-B/32 MB_READER_SAMPLES=8000 MB_PIPE_PER_SHARD=4096 timeout 1200 python tools/microbench.py reader pipeline > gpurun_out/r3j_reader_pipeline_b32.log 2>&1; echo "mb rc=$?"
grep "WebdatasetReader\|host cores\|tar iteration\|pipeline worker" gpurun_out/r3j_reader_pipeline_b32.log
