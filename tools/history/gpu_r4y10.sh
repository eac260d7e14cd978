#!/bin/bash
# round 4: two-level threshold of the int8 first stage for batches of more than 64 queries -- tests, then the kNN legs
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_knn_gpu.py -m gpu -q -x -k "rq or i8" > gpurun_out/r04y10_i8_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r04y10_i8_tests.log | cut -c1-300
( timeout 600 python bench.py --steps 2 --warmup 1 --no-parity --no-ab --cpu-seconds 0 --knn-batches 64,128,256 ) 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['knn']
print('planted', d.get('planted_neighbour_top1'), 'fallbacks', d.get('wide_fallbacks'))
for b in d['by_batch']: print(b['B'], b['qps'], b['ms_per_batch'], b.get('scan_ms'), b.get('passes_over_hbm'), b.get('proof_failures'))
" > gpurun_out/r04y10_knn_two_level.log 2>&1
cat gpurun_out/r04y10_knn_two_level.log
