#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05m_pytest_gpu.log 2>&1; tail -6 gpurun_out/r05m_pytest_gpu.log
