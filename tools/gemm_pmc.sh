#!/bin/bash
# MFMA utilisation of the persistent GEMM from hardware counters (separate --pmc pass, kernel-trace only), driven by the
# Python-free tools/gemm_bench:   bash tools/gemm_pmc.sh <tag> [shape ...]      (shapes "M,N,K,epi", default QKV + fc1 + fc2)
set -u
TAG=${1:-rX}; shift
SHAPES=${@:-"65536,3072,1024,0 65536,4096,1024,1 65536,1024,4096,3 65536,1024,1024,3"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_gemm_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT -o p -- $ROOT/tools/gemm_bench -r 3 $SHAPES -- 3 > $OUT/run.log 2>&1
python3 - <<PY | tee $OUT/summary.txt
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    if 'gemm256sp' not in r['Kernel_Name']: continue
    key = (r['Kernel_Name'][:40], r['Grid_Size'], r.get('Dispatch_Id', ''))
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
# one line per dispatch: MFMA busy cycles per SIMD / GUI-active cycles
seen = collections.defaultdict(list)
for (name, grid, did), c in agg.items():
    gui = sum(c['GRBM_GUI_ACTIVE']) / max(len(c['GRBM_GUI_ACTIVE']), 1)
    mf = sum(c['SQ_VALU_MFMA_BUSY_CYCLES'])
    seen[name].append((gui, mf))
for name, v in seen.items():
    for gui, mf in v[:12]:
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is the busy-cycle total of all
        # 1024 SIMDs (= MFMAs x 32 for v_mfma_f32_32x32x16_bf16): utilisation = busy / (1024 x kernel cycles)
        cyc = gui / 8.0
        print(f"{name}  kernel cycles {cyc:9.0f}  MFMA busy (all SIMDs) {mf:13.0f}  MFMA utilisation {100.0 * mf / (1024.0 * max(cyc, 1)):5.1f} %")
PY
