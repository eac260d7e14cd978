#!/bin/bash
# one-off measurement batch (round 2): GEMM prefetch A/B, vendor-GEMM calibration, IVF build at 20 M x 1024, reader rates
set -u
TAG=${1:-r02c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
{
  echo "== gemm prefetch A/B (cfg = variant:dbg:flags; flags = raster | 8 boundary prefetch | distance << 4; dbg 32 = register-sink prefetch in the main loop)"
  CLIPX_LIB=libclipx_ablate.so GEMM_BENCH_SHADOW=1 timeout 300 tools/gemm_bench -r 12 65536,3072,1024,0 65536,4096,1024,1 65536,1024,1024,3 65536,1024,4096,3 -- 3:0:2 3:0:10 3:0:98 3:0:106 3:0:138 3:32:2 3:32:10 3:16:2 3:16:10
} > $OUT/${TAG}_gemm_prefetch.log 2>&1
tail -60 $OUT/${TAG}_gemm_prefetch.log
timeout 300 python tools/calib_blas.py > $OUT/${TAG}_vendor_gemm.log 2>&1; cat $OUT/${TAG}_vendor_gemm.log
free -g | head -2
MB_IVF_ROWS=20000000 MB_IVF_D=1024 MB_IVF_NLIST=16384 MB_IVF_RECALL_ROWS=10000000 timeout 900 python tools/microbench.py ivf > $OUT/${TAG}_ivf20m.log 2>&1; cat $OUT/${TAG}_ivf20m.log
timeout 400 python tools/microbench.py reader > $OUT/${TAG}_reader.log 2>&1; tail -12 $OUT/${TAG}_reader.log
