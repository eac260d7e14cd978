#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3f_pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r3f_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r3f_bench.log 2>gpurun_out/r3f_bench.err; echo "bench rc=$?"
tail -2 gpurun_out/r3f_bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3f_bench.log') if x.startswith('{')][-1]
j=json.loads(l)
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['ms_per_step'], j['kernel_ms_per_step'], j['parity'], j['cpu_baseline'])
print([(r['B'], r['qps'], r['ms_per_batch'], r['scan_ms']) for r in j['knn']['by_batch']])
PY
