#!/bin/bash
# One GPU visit: parity tests, the default bench line, a rocprofv3 kernel-trace of the same command, microbench.
# usage (on the GPU box, from the repo root): bash tools/gpu_round.sh <tag> [quick]
set -u
TAG=${1:-rX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
tail -3 $OUT/pytest_gpu_$TAG.log
timeout 600 python tools/microbench.py > $OUT/microbench_$TAG.log 2>&1; cat $OUT/microbench_$TAG.log
if [ "${2:-}" != "quick" ]; then
  ( time timeout 900 python bench.py ) > $OUT/bench_$TAG.log 2>&1; tail -5 $OUT/bench_$TAG.log
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $ROOT/bench.py --cpu-seconds 0 --no-parity > $OUT/rocprof_$TAG.log 2>&1
  ls -R $OUT/prof_$TAG | head
fi
