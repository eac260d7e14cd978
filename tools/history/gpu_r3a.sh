#!/bin/bash
# round-3 GPU visit A: parity tests, per-shape GEMM numbers (fp16 residual stream vs the f32 epilogue), the bench line,
# BASELINE config 5 small then at size.   usage (GPU box, repo root): bash tools/gpu_r3a.sh <tag>
set -u
TAG=${1:-r03a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
tail -15 $OUT/pytest_gpu_$TAG.log
timeout 300 tools/gemm_bench -r 10 65792,3072,1024,16 65792,3072,1024,0 65792,4096,1024,17 65792,4096,1024,1 65792,1024,1024,6 65792,1024,1024,3 65792,1024,4096,6 65792,1024,4096,3 19712,2304,768,16 19712,768,768,6 19712,768,3072,6 -- 3 > $OUT/gemm_bench_$TAG.log 2>&1; cat $OUT/gemm_bench_$TAG.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_$TAG.log 2>&1; tail -4 $OUT/bench_$TAG.log | cut -c1-3000
timeout 600 python tools/config5.py --rows 20000000 --nlist 16384 --seconds 1 --threads 1,64 > $OUT/config5_small_$TAG.log 2>&1; rc=$?; tail -30 $OUT/config5_small_$TAG.log | cut -c1-400
if [ $rc -eq 0 ]; then
  timeout 900 python tools/config5.py > $OUT/config5_$TAG.log 2>&1; echo "config5 rc=$?"; tail -40 $OUT/config5_$TAG.log | cut -c1-400
fi
