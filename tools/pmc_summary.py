#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv: launches, mean duration, effective clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)), SQ busy,
and FETCH_SIZE / WRITE_SIZE totals when present.   usage: pmc_summary.py <csv> [name-substring ...]"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
        continue
    key = (name, r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[key] = ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("VGPR_Count"), r.get("Accum_VGPR_Count"),
                 r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
agg = collections.defaultdict(list)
for key, c in rows.items():
    agg[key[0]].append((meta[key], c))
for name, v in sorted(agg.items(), key=lambda kv: -sum(m[0] for m, _ in kv[1])):
    n = len(v)
    us = sum(m[0] for m, _ in v) / n
    cs = collections.defaultdict(float)
    for _, c in v:
        for k, x in c.items():
            cs[k] += x / n
    line = f"{name[:110]}\n    launches {n}  mean {us:.1f} us  vgpr/agpr/lds/grid/wg {v[0][0][1:]}"
    if "GRBM_GUI_ACTIVE" in cs:
        cyc = cs["GRBM_GUI_ACTIVE"] / 8.0
        line += f"\n    kernel cycles {cyc:.0f}  effective clock {cyc / us / 1e3:.3f} GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
            line += f"  mfma_busy_frac {cs['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * max(cyc, 1)):.3f}"
        if "SQ_BUSY_CYCLES" in cs:
            line += f"  SQ_BUSY_CYCLES {cs['SQ_BUSY_CYCLES']:.4g}"
    for k in ("FETCH_SIZE", "WRITE_SIZE"):
        if k in cs:
            line += f"\n    {k} {cs[k] / 1e6:.1f} MB per launch (raw; x2 for 16-B/lane streaming reads on gfx950)"
    other = {k: x for k, x in cs.items() if k not in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "FETCH_SIZE", "WRITE_SIZE")}
    if other:
        line += "\n    " + "  ".join(f"{k}={x:.4g}" for k, x in sorted(other.items()))
    print(line)
