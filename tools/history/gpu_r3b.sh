#!/bin/bash
# round-3 GPU visit B: the evidence run -- parity tests, the bench line with the config-5 leg, rocprofv3 kernel stats of the
# same command, the two PMC passes the `traffic` fields come from, the request bench.
#   usage (GPU box, repo root): bash tools/gpu_r3b.sh <tag>
set -u
TAG=${1:-r03b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
tail -4 $OUT/pytest_gpu_$TAG.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 --ivf ) > $OUT/bench_$TAG.log 2>&1; tail -4 $OUT/bench_$TAG.log | cut -c1-600
timeout 600 python tools/request_bench.py > $OUT/request_$TAG.log 2>&1; tail -16 $OUT/request_$TAG.log | cut -c1-160
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $ROOT/bench.py --cpu-seconds 0 --no-parity > $OUT/rocprof_$TAG.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_$TAG -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --knn-scans 2 --cpu-seconds 0 --no-parity > $OUT/pmc_${c}_$TAG.log 2>&1
done
cd $ROOT
python3 tools/traffic_summary.py $OUT/pmc_FETCH_SIZE_$TAG/p_counter_collection.csv $OUT/pmc_WRITE_SIZE_$TAG/p_counter_collection.csv --steps 3 --warmup 1 > $OUT/traffic_$TAG.json; head -c 1500 $OUT/traffic_$TAG.json
find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -2
# keep the merge-back small: drop the raw traces, keep stats + counter csvs
find $OUT/prof_$TAG -name "*kernel_trace.csv" -delete
