#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_knn_gpu.py -m gpu -q -x -k "rq or i8" > gpurun_out/r04y6_i8_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04y6_i8_tests.log | cut -c1-300
run() { ( timeout 900 python bench.py --steps 2 --warmup 1 --no-parity --no-ab --cpu-seconds 0 --knn-batches 1,32,64,128,256 ) 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['knn']
print('planted', d.get('planted_neighbour_top1'), 'fallbacks', d.get('wide_fallbacks'), 'i8', d.get('int8_first_stage'))
for b in d['by_batch']: print(b['B'], b['qps'], b['ms_per_batch'], b.get('scan_ms'), b.get('hbm_frac'), b.get('proof_failures'), b['roofline']['mfma_frac'], b.get('int8_first_stage'))
"; }
{ echo "== int8 first stage"; run; } > gpurun_out/r04y6_knn_i8.log 2>&1
cat gpurun_out/r04y6_knn_i8.log
