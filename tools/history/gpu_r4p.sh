#!/bin/bash
# round 4: RQ scan, 8 waves x 32 queries against 4 waves x 64 queries (KNNX_RQ_4X64=1), both on 16x16x32; same box, alternating
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { ( timeout 600 python bench.py --steps 2 --warmup 1 --no-parity --no-ab --cpu-seconds 0 --knn-batches 256 ) 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['knn']
for b in d['by_batch']: print(b['B'], b['qps'], b['ms_per_batch'], b.get('scan_ms'), b.get('hbm_frac'), b.get('proof_failures'), b.get('roofline',{}).get('mfma_frac'))
"; }
{
echo "== 8 x 32"; run
echo "== 4 x 64"; KNNX_RQ_4X64=1 run
echo "== 8 x 32"; run
echo "== 4 x 64"; KNNX_RQ_4X64=1 run
} > gpurun_out/r04p_rq_cfg.log 2>&1
cat gpurun_out/r04p_rq_cfg.log
