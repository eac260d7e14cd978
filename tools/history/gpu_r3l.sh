#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q -m gpu > gpurun_out/r3l_knn_tests.log 2>&1; echo "knn tests rc=$?"; tail -3 gpurun_out/r3l_knn_tests.log
timeout 600 python bench.py > gpurun_out/r3l_bench.log 2>gpurun_out/r3l_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r3l_bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3l_bench.log') if x.startswith('{')][-1]
j=json.loads(l)
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['kernel_ms_per_step'])
print([(r['B'], r['qps'], r['ms_per_batch'], r['scan_ms']) for r in j['knn']['by_batch']])
print(j['knn']['checks'])
PY
