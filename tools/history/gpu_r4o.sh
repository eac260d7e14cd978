#!/bin/bash
# round 4: RQ scan / assignment kernel on v_mfma_f32_16x16x32_f16 -- kNN + service GPU tests, then the kNN legs of the bench
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_knn_gpu.py tests/test_service_gpu.py -m gpu -q -x > gpurun_out/r04o_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04o_pytest.log
( time timeout 600 python bench.py --steps 3 --warmup 1 --no-parity --no-ab --cpu-seconds 0 --knn-batches 64,256 ) > gpurun_out/r04o_bench.log 2>&1
grep '^{' gpurun_out/r04o_bench.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['knn']
print({k:v for k,v in d.items() if k not in ('by_batch',)})
for b in d['by_batch']: print(b['B'], b['qps'], b['ms_per_batch'], b.get('scan_ms'), b.get('hbm_frac'), b.get('proof_failures'), b.get('roofline',{}).get('mfma_frac'))
"
