#!/bin/bash
# PMC of the 256x256 GEMM variants at the QKV shape: usage gemm_pmc.sh <tag> "<variant:dbg> ..."
set -u
TAG=${1:-rX}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp MB_GEMM=${MB_GEMM:-qkv} MB_REPS=3
cd /tmp
for vd in "$@"; do
  v=${vd%%:*}; d=${vd##*:}
  MB_VARIANTS=$v CLIPX_GEMM_DBG=$d timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/v${v}d${d} -o p -- python $ROOT/tools/microbench.py gemm > $OUT/v${v}d${d}.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
for f in sorted(glob.glob("$OUT/*/p_counter_collection.csv")):
    agg=collections.defaultdict(list); dur=[]
    for r in csv.DictReader(open(f)):
        if 'gemm256' not in r['Kernel_Name']: continue
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
        dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print(os.path.basename(os.path.dirname(f)), 'us=%.1f'%(sum(dur)/max(len(dur),1)), ' '.join('%s=%.4g'%(k,sum(v)/len(v)) for k,v in sorted(agg.items())))
PY
