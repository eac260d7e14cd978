#!/bin/bash
# PMC of one microbench target: usage pmc.sh <tag> <kernel-substring> <microbench args...>
set -u
TAG=$1; KSUB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp MB_REPS=3
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -o p -- python $ROOT/tools/microbench.py "$@" > $OUT/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_WAVES --output-format csv -d $OUT/b -o p -- python $ROOT/tools/microbench.py "$@" > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections, os
for f in sorted(glob.glob("$OUT/*/p_counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list); res={}
    for r in csv.DictReader(open(f)):
        if "$KSUB" not in r['Kernel_Name']: continue
        k=r['Kernel_Name'][:70]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
        res[k]=(r['VGPR_Count'],r['Accum_VGPR_Count'],r['LDS_Block_Size'],r['Grid_Size'],r['Workgroup_Size'])
    for k in agg:
        print(k, 'us=%.1f'%(sum(dur[k])/len(dur[k])), 'vgpr/agpr/lds/grid/wg=',res[k])
        print('   ',' '.join('%s=%.4g'%(c,sum(v)/len(v)) for c,v in sorted(agg[k].items())))
PY
