#!/bin/bash
# A/B of tile rasters of gemm256sp (ablation build, CLIPX_GEMM_FLAGS = raster id) on the ViT-L/14 bs=256 QKV and fc1 shapes:
# interleaved timing with the phase timer, then one rocprofv3 --pmc FETCH_SIZE pass per raster (kernel-trace only) --
# L2 -> fabric read bytes per launch (x2: gfx950 counts 128-B requests as 64 B).   usage: bash tools/gemm_raster.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export CLIPX_LIB=libclipx_ablate.so TMPDIR=/tmp
LOG=$OUT/${TAG}_gemm_raster.log
cd $ROOT
{
  echo "== timing (median of 10 interleaved launches; cfg = variant:dbg:raster; 16 = phase timer)"
  timeout 300 tools/gemm_bench -r 10 65536,3072,1024,0 65536,4096,1024,1 65536,1024,1024,3 -- 3:0:0 3:0:1 3:0:2 3:16:0 3:16:1 3:16:2
  for r in 0 1 2; do
    cd /tmp
    timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_raster_$r -o p -- $ROOT/tools/gemm_bench -r 3 65536,3072,1024,0 -- 3:0:$r > /dev/null 2>&1
    cd $ROOT
    python3 - $OUT/pmc_raster_$r/p_counter_collection.csv $r <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "gemm256sp_kernel" in r["Kernel_Name"]]
print(f"== raster {sys.argv[2]}: QKV 65536x3072x1024, gemm256sp launches {len(v)}, FETCH_SIZE x2 = {2 * 1024 * sum(v) / max(len(v), 1) / 1e6:.1f} MB per launch "
      f"(algorithmic operand reads 140.5 MB; output-stationary bound of an 8 x 4 tile XCD round: 576 MB)")
PY
  done
} > $LOG 2>&1
cat $LOG
