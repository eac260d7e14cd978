#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python tools/config5.py --threads 1,64 --seconds 1 > gpurun_out/r05l_config5_balanced.log 2>&1
grep -v "^CONFIG5" gpurun_out/r05l_config5_balanced.log | tail -40
