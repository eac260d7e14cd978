#!/bin/bash
# one-off (round 2): row-scale touch A/B in the bf16-output GEMM epilogues; names of the vendor GEMM kernels torch picks
set -u
TAG=${1:-r02d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
{
  echo "== row-scale touch A/B (cfg = variant:dbg:flags; flags 2 = product, 10 = without the touch; dbg 16 = phase timer)"
  CLIPX_LIB=libclipx_ablate.so timeout 300 tools/gemm_bench -r 16 65536,3072,1024,0 65536,4096,1024,1 -- 3:0:2 3:0:10 3:16:2 3:16:10
} > $OUT/${TAG}_gemm_rowscale_touch.log 2>&1
cat $OUT/${TAG}_gemm_rowscale_touch.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vendor_prof -o v -- python $ROOT/tools/calib_blas.py > $OUT/${TAG}_vendor_gemm2.log 2>&1
cd $ROOT
python3 - <<'PY' > $OUT/${TAG}_vendor_kernels.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/vendor_prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:400], r["Calls"], r["AverageNs"])
PY
cat $OUT/${TAG}_vendor_kernels.txt
