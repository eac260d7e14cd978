#!/bin/bash
# round 4: kNN legs of the bench with the int8 first stage (default) and without (KNNX_I8=0), same box
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { ( timeout 900 python bench.py --steps 2 --warmup 1 --no-parity --no-ab --cpu-seconds 0 --knn-batches 1,32,64,256 ) 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['knn']
print('planted', d.get('planted_neighbour_top1'), 'fallbacks', d.get('wide_fallbacks'), 'i8', d.get('int8_first_stage'))
for b in d['by_batch']: print(b['B'], b['qps'], b['ms_per_batch'], b.get('scan_ms'), b.get('hbm_frac'), b.get('proof_failures'), b.get('roofline',{}).get('kernel'))
"; }
{
echo "== int8 first stage"; run
echo "== KNNX_I8=0"; KNNX_I8=0 run
} > gpurun_out/r04y_knn_i8.log 2>&1
cat gpurun_out/r04y_knn_i8.log
