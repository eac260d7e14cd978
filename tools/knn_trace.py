"""Kernel-level timeline of one flat-index search batch (for `rocprofv3 --kernel-trace`): fills a synthetic index, runs a few
batches of B queries, and -- when given the trace CSV of a previous run of itself -- prints the kernels of the last batch with their
durations and the gaps between them.    python tools/knn_trace.py run  [rows] [B]      |      python tools/knn_trace.py show trace.csv"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rows, B):
    import numpy as np

    from clip_retrieval_amd.knn import Mi355xIndex
    ix = Mi355xIndex(768)
    ix.synth_fill(rows, 7)
    q = np.random.default_rng(3).standard_normal((B, 768)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    import time

    for _ in range(3):
        ix.search(q, 40)
    t0 = time.perf_counter()
    for _ in range(5):
        ix.search(q, 40)
    print(f"B={B}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per batch, {B * 5 / (time.perf_counter() - t0):.1f} QPS", flush=True)
    if os.environ.get("KNN_TRACE_SAVE"):  # the ids / scores of the batch, for a comparison between two settings of a switch
        D, I = ix.search(q, 40)
        np.save(os.environ["KNN_TRACE_SAVE"] + "_I.npy", I)
        np.save(os.environ["KNN_TRACE_SAVE"] + "_D.npy", D)
        print("stats (proof-served, fallbacks):", ix.stats(), flush=True)
    ix.close()


def show(path, last=60):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    prev = None
    for r in rows[-last:]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (st - prev) / 1e3 if prev else 0.0
        prev = en
        print("%-70s %9.1f us  gap %7.1f  grid %s" % (r["Kernel_Name"][:70], (en - st) / 1e3, gap, r.get("Grid_Size_X", r.get("Grid_Size"))))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000, int(sys.argv[3]) if len(sys.argv) > 3 else 256)
    else:
        show(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
