#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
for v in new old; do
  if [ $v = old ]; then export CLIPX_LIB=libclipx_ablate.so CLIPX_ATTN_CFG=10; fi
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_attn_$v -o p -- $R/tools/attn_bench 256 257 16 64 0 > $R/gpurun_out/pmc_attn_$v.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for v in ("new","old"):
    f=glob.glob(f"gpurun_out/pmc_attn_{v}/*counter_collection.csv")
    if not f: print(v,"no csv"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'attention' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(v, {k: round(sum(x)/len(x)) for k,x in sorted(agg.items())})
PY
