#!/bin/bash
# Ablation + PMC of the persistent 256x256 GEMM at the ViT-L/14 QKV shape (65792 x 3072 x 1024, bf16 out).
set -u
TAG=${1:-rX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp MB_VARIANTS=2 MB_GEMM=qkv MB_REPS=10
for d in 0 1 2 3; do
  echo "== CLIPX_GEMM_DBG=$d" ; CLIPX_GEMM_DBG=$d timeout 120 python tools/microbench.py gemm 2>&1 | grep gemm
done | tee $OUT/gemm_ablate_$TAG.log
cd /tmp
export MB_REPS=3
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_gemm_$TAG/sq -o sq -- python $ROOT/tools/microbench.py gemm > $OUT/pmc_gemm_${TAG}_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_gemm_$TAG/tcc -o tcc -- python $ROOT/tools/microbench.py gemm > $OUT/pmc_gemm_${TAG}_tcc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/pmc_gemm_$TAG/sq2 -o sq2 -- python $ROOT/tools/microbench.py gemm > $OUT/pmc_gemm_${TAG}_sq2.log 2>&1
ls -R $OUT/pmc_gemm_$TAG | head -30
