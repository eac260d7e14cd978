#!/bin/bash
# One GPU visit: parity tests, microbench, the default bench line, a rocprofv3 kernel-trace of the same command and
# the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) that bench.py's `traffic` fields
# come from.   usage (on the GPU box, from the repo root): bash tools/gpu_round.sh <tag> [quick|lite]
# quick = tests + microbench; lite = tests + bench line; default = everything
set -u
TAG=${1:-rX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
# prof = only the profiling passes (kernel stats + the three counter passes) of the bench command
if [ "${2:-}" != "prof" ]; then
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
tail -3 $OUT/pytest_gpu_$TAG.log
fi
if [ "${2:-}" = "lite" ]; then
  ( time timeout 900 python bench.py ) > $OUT/bench_$TAG.log 2>&1; tail -5 $OUT/bench_$TAG.log
  exit 0
fi
if [ "${2:-}" != "prof" ]; then
MB_VARIANTS=1,3 MB_KNN_ROWS=100000000 timeout 900 python tools/microbench.py gemm attn ln knn b1 ivf e2e reader pipeline > $OUT/microbench_$TAG.log 2>&1; cat $OUT/microbench_$TAG.log
fi
if [ "${2:-}" != "quick" ]; then
  if [ "${2:-}" != "prof" ]; then ( time timeout 900 python bench.py ) > $OUT/bench_$TAG.log 2>&1; tail -5 $OUT/bench_$TAG.log; fi
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $ROOT/bench.py --profile-run > $OUT/rocprof_$TAG.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_$TAG -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --knn-scans 2 --profile-run > $OUT/pmc_${c}_$TAG.log 2>&1
  done
  timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_MFMA_$TAG -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --knn-scans 2 --profile-run > $OUT/pmc_MFMA_$TAG.log 2>&1
  cd $ROOT
  python3 tools/mfma_busy_summary.py $(find $OUT/pmc_MFMA_$TAG -name 'p_counter_collection.csv' | head -1) > $OUT/mfma_busy_$TAG.json; head -c 1500 $OUT/mfma_busy_$TAG.json
  python3 tools/traffic_summary.py $(find $OUT/pmc_FETCH_SIZE_$TAG -name 'p_counter_collection.csv' | head -1) $(find $OUT/pmc_WRITE_SIZE_$TAG -name 'p_counter_collection.csv' | head -1) --steps 3 --warmup 1 > $OUT/traffic_$TAG.json; cat $OUT/traffic_$TAG.json
  ls -R $OUT/prof_$TAG | head
fi
