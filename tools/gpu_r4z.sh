#!/bin/bash
# round-4 evidence run: full GPU suite, the default bench line (+ --ivf: BASELINE config 5's shard at its size), smoke, a rocprofv3
# kernel-trace of the same command, the two PMC passes behind `roofline.traffic`.   usage: bash tools/gpu_r4z.sh <tag> [noivf]
set -u
TAG=${1:-r04z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu_$TAG.log | tail -12
IVF="--ivf"; [ "${2:-}" = "noivf" ] && IVF=""
( time timeout 1500 python bench.py $IVF ) > $OUT/bench_$TAG.log 2>&1; tail -4 $OUT/bench_$TAG.log | cut -c1-1500
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $ROOT/bench.py --cpu-seconds 0 --no-parity --no-ab > $OUT/rocprof_$TAG.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_$TAG -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --knn-scans 2 --cpu-seconds 0 --no-parity --no-ab > $OUT/pmc_${c}_$TAG.log 2>&1
done
cd $ROOT
python3 tools/traffic_summary.py $(find $OUT/pmc_FETCH_SIZE_$TAG -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_WRITE_SIZE_$TAG -name "*counter_collection.csv" | head -1) --steps 3 --warmup 1 > $OUT/traffic_$TAG.json; cat $OUT/traffic_$TAG.json | cut -c1-600
cp $(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$TAG.csv
head -12 $OUT/kernel_stats_$TAG.csv | cut -c1-160
find $OUT/prof_$TAG $OUT/pmc_FETCH_SIZE_$TAG $OUT/pmc_WRITE_SIZE_$TAG -name "*kernel_trace.csv" -delete 2>/dev/null
find $OUT/pmc_FETCH_SIZE_$TAG $OUT/pmc_WRITE_SIZE_$TAG -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
du -sh $OUT | tail -1
