#!/usr/bin/env python3
"""MFMA utilisation from ONE rocprofv3 --pmc pass (GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES, kernel-trace only) of
bench.py: per kernel INSTANTIATION (template arguments kept) the launches, mean duration, effective shader clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration) and mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); per FAMILY the
time-weighted mean (what bench.py puts beside the arithmetic fraction in `roofline`).  north_star: "rocprof reports ... MFMA
utilisation for the ViT GEMMs against gfx950 peak".   usage: mfma_busy_summary.py counter_collection.csv -> JSON"""
import collections
import csv
import json
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from source_digest import source_digest  # noqa: E402

FAMILIES = {"gemm256sp_kernel": "gemm", "gemm256w4_kernel": "gemm", "gemm_bf16_kernel": "gemm", "attention_pk_kernel": "attention", "attention_kernel": "attention",
            "knn_rq8_scan_kernel": "knn_rq8_scan_kernel", "knn_rq_scan_kernel": "knn_rq_scan_kernel", "knn_scan_kernel": "knn_scan_kernel",
            "knn_assign_kernel": "knn_assign_kernel"}
rows = collections.defaultdict(lambda: collections.defaultdict(float))
dur = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if not any(k in name for k in FAMILIES):
        continue
    key = (name, r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
    dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def short(name):
    m = re.search(r"(\w+_kernel)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2)}>" if m else name[:60]


inst = collections.defaultdict(list)
for key, c in rows.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0:
        continue
    inst[short(key[0])].append((dur[key], cyc, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)))
out = {"instantiations": {}, "families": {}}
fam = collections.defaultdict(list)
MIN_US = 100.0  # GRBM_GUI_ACTIVE of a dispatch includes a few microseconds of dispatch overhead: below ~100 us it inflates the cycle
                # count (a 13 us kernel reads as 3.8 "GHz") and deflates the fraction -- such launches are listed but not aggregated
for name, v in sorted(inst.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    us = sum(x[0] for x in v)
    long = us / len(v) >= MIN_US
    out["instantiations"][name] = {"launches": len(v), "mean_us": round(us / len(v), 1),
                                   "effective_clock_ghz": round(sum(x[1] for x in v) / us / 1e3, 3) if long else None,
                                   "mfma_busy_frac": round(sum(x[0] * x[2] for x in v) / us, 4) if long else None}
    if not long:
        continue
    for k, f in FAMILIES.items():
        if k in name:
            fam[f] += v
            break
for f, v in fam.items():
    us = sum(x[0] for x in v)
    out["families"][f] = {"launches": len(v), "total_ms": round(us / 1e3, 3), "effective_clock_ghz": round(sum(x[1] for x in v) / us / 1e3, 3),
                          "mfma_busy_frac": round(sum(x[0] * x[2] for x in v) / us, 4)}
out["note"] = ("SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the 1024 SIMDs; kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter "
               "comes back summed over the 8 XCDs); profiled passes clock ~3 % lower than un-profiled ones (MI355X_MICROARCH.md, DVFS): compare fractions, not times")
out["source_digest"] = source_digest()
json.dump(out, sys.stdout, indent=1)
