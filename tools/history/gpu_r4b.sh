#!/bin/bash
# round 4, second GPU visit: search-side changes (IVF from disk, k = 100 000, native coalescer + fused dedup, violence filter)
# under the GPU tests, then the served-request leg of config 5 on a 30 M-row shard (full size costs 3 GPU-minutes of build)
set -u
TAG=${1:-r04b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_knn_gpu.py tests/test_service_gpu.py tests/test_service.py tests/test_clip_gpu.py -m gpu -q -k "not test_gemm_epilogues" > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
grep -E "passed|failed|^FAILED|^ERROR|^E  " $OUT/pytest_gpu_$TAG.log | tail -30
timeout 900 python tools/config5.py --rows 30000000 --nlist 16384 --nprobe 16 --threads 1,8,64 --seconds 3 > $OUT/config5_$TAG.log 2>&1; grep -v "^CONFIG5" $OUT/config5_$TAG.log | tail -25
