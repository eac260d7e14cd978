"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch).

Correction prescribed by /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read (16 B/lane, global_load and LDS-DMA alike), so the read side is
doubled for the kernels whose reads are of that kind (the kNN scan, the GEMMs' LDS-DMA); WRITE_SIZE is uncalibrated
and reported as is.  Output: JSON {kernel: {launches, fetch_bytes_per_launch (corrected), write_bytes_per_launch}}."""
import collections
import csv
import json
import sys

GROUPS = {"knn_scan_kernel": "knn_scan_kernel", "gemm256sp_kernel": "gemm", "gemm256_kernel": "gemm", "gemm_bf16_kernel": "gemm",
          "attention_kernel": "attention_kernel", "layernorm_kernel": "layernorm_kernel"}


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        for sub, name in GROUPS.items():
            if sub in r["Kernel_Name"]:
                agg[name].append(float(r["Counter_Value"]))
                break
    return agg


def drop_gated(agg):
    """The wide kNN scan launches two gated exact scans per batch that exit at once (no traffic) unless a proof failed:
    keep only the launches that actually streamed the index."""
    v = agg.get("knn_scan_kernel")
    if v:
        top = max(v)
        agg["knn_scan_kernel"] = [x for x in v if x >= 0.5 * top]
    return agg


fetch, write = drop_gated(load(sys.argv[1])), load(sys.argv[2])
if "knn_scan_kernel" in write and "knn_scan_kernel" in fetch:
    write["knn_scan_kernel"] = sorted(write["knn_scan_kernel"], reverse=True)[: len(fetch["knn_scan_kernel"])]
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    out[k] = {"launches": len(f), "fetch_bytes_per_launch": round(2 * 1024 * sum(f) / max(len(f), 1)),
              "fetch_kb_raw_per_launch": round(sum(f) / max(len(f), 1), 1),
              "write_bytes_per_launch": round(1024 * sum(w) / max(len(w), 1))}
json.dump({"note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncorrected",
           "kernels": out}, sys.stdout, indent=1)
