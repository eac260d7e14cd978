#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python tools/rq8_phases.py > gpurun_out/r05o_rq8_phases.log 2>&1; cat gpurun_out/r05o_rq8_phases.log
