#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/mfma_probe2 > gpurun_out/r04j_mfma_probe2.log 2>&1; cat gpurun_out/r04j_mfma_probe2.log
