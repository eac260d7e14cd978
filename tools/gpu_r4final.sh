#!/bin/bash
# round 4, final tree: the full GPU suite, smoke, the default bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r04final_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04final_pytest_gpu.log; tail -3 gpurun_out/r04final_pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04final_smoke.log 2>&1; tail -1 gpurun_out/r04final_smoke.log
( time timeout 900 python bench.py ) > gpurun_out/r04final_bench.log 2>&1; grep '^{' gpurun_out/r04final_bench.log | tail -1 | cut -c1-400
