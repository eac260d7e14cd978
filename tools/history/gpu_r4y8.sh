#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python bench.py --steps 2 --warmup 1 --no-parity --no-ab --cpu-seconds 0 --knn-batches 1,64,128,256 ) 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['knn']
print('planted', d.get('planted_neighbour_top1'), 'fallbacks', d.get('wide_fallbacks'))
for b in d['by_batch']: print(b['B'], b['qps'], b['ms_per_batch'], b.get('scan_ms'), b.get('hbm_frac'))
" > gpurun_out/r04y8_knn_ring6.log 2>&1
cat gpurun_out/r04y8_knn_ring6.log
