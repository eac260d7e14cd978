"""HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch) of bench.py.

Correction prescribed by /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read (16 B/lane, global_load and LDS-DMA alike), so the read side is
doubled for the kernels whose reads are of that kind (the kNN scans, the GEMMs' LDS-DMA); WRITE_SIZE is uncalibrated
and reported as is.

One denominator per kernel family:
  encode kernels (gemm, attention, layernorm)  bytes per STEP = sum over all launches / number of step-equivalents the
                                                profiled command ran (warm-up + timed + the extra per-kind pass + the
                                                image-only and text-only passes, which add up to one step per pair)
  kNN scans                                     bytes per full pass over the index (launches that streamed < half of the
                                                largest launch -- gated fallbacks, the strided sample passes -- are dropped)
usage: traffic_summary.py FETCH.csv WRITE.csv --steps K --warmup W  ->  JSON {kernels: {name: {bytes_per_unit, unit, ...}}}"""
import argparse
import collections
import csv
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from source_digest import source_digest  # noqa: E402

GROUPS = {"knn_rq8_scan_kernel": "knn_rq8_scan_kernel", "knn_rq_scan_kernel": "knn_rq_scan_kernel", "knn_scan_kernel": "knn_scan_kernel", "gemm256sp_kernel": "gemm", "gemm256w4_kernel": "gemm", "gemm_bf16_kernel": "gemm",
          "attention_kernel": "attention", "attention_pk_kernel": "attention", "layernorm_kernel": "layernorm"}
SCANS = ("knn_scan_kernel", "knn_rq_scan_kernel", "knn_rq8_scan_kernel")


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        for sub, name in GROUPS.items():
            if sub in r["Kernel_Name"]:
                agg[name].append(float(r["Counter_Value"]))
                break
    return agg


def full_scans(v):
    top = max(v) if v else 0.0
    return [x for x in v if x >= 0.5 * top]


ap = argparse.ArgumentParser()
ap.add_argument("fetch")
ap.add_argument("write")
ap.add_argument("--steps", type=int, required=True)
ap.add_argument("--warmup", type=int, required=True)
a = ap.parse_args()
half = max(2, a.steps // 2)
step_equiv = a.warmup + a.steps + half + half  # + extra per-kind pass + (image-only + text-only passes = `half` whole steps)
fetch, write = load(a.fetch), load(a.write)
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    if k in SCANS:
        f = full_scans(f)
        w = sorted(w, reverse=True)[: len(f)]
        units, unit = max(len(f), 1), "pass over the index"
    else:
        units, unit = step_equiv, "step"
    fb, wb = 2 * 1024 * sum(f), 1024 * sum(w)
    out[k] = {"unit": unit, "units": units, "launches": len(f), "fetch_bytes_per_unit": round(fb / units), "write_bytes_per_unit": round(wb / units),
              "bytes_per_unit": round((fb + wb) / units)}
json.dump({"note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncorrected; "
                   "counts are L2 <-> fabric requests (Infinity Cache hits included)",
           "command": f"bench.py --steps {a.steps} --warmup {a.warmup}", "source_digest": source_digest(), "kernels": out}, sys.stdout, indent=1)
