#!/bin/bash
# round 4, first GPU visit: the encode-side changes (parity gate, fp16 q/k/v, range guard, device normalisation) under the GPU tests,
# then the bench line (all 256 rows gated against the stored oracle rows)
set -u
TAG=${1:-r04a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_clip_gpu.py tests/test_reader_reference_tensors.py tests/test_service_gpu.py tests/test_preprocess_gpu.py -m gpu -q -s > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
grep -E "passed|failed|outliers|^FAILED|^ERROR" $OUT/pytest_gpu_$TAG.log | tail -30
( time timeout 900 python bench.py ) > $OUT/bench_$TAG.log 2>&1; tail -4 $OUT/bench_$TAG.log | cut -c1-3000
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log
