#!/usr/bin/env python3
"""B = 256 flat search over 100 M x 768 for several workgroup counts of the RQ path's threshold-sample scans (KNNX_RQ_SAMPLE_GRID,
read at index creation).  usage: python tools/rq_sample_grid.py [grid ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from clip_retrieval_amd.knn import Mi355xIndex
from clip_retrieval_amd.synth import perturbed_queries

grids = [int(a) for a in sys.argv[1:]] or [256, 128, 64, 32]
ref = None
for g in grids:
    os.environ["KNNX_RQ_SAMPLE_GRID"] = str(g)
    ix = Mi355xIndex(768)
    ix.synth_fill(int(os.environ.get("ROWS", "100000000")), 3)
    rng = np.random.default_rng(0)
    planted = np.sort(rng.choice(ix.ntotal, 256, replace=False))
    q = perturbed_queries(ix.reconstruct_batch(planted))
    D, I = ix.search(q, 40)
    if ref is None:
        ref = (D, I)
    same = bool(np.array_equal(I, ref[1]) and np.array_equal(D, ref[0]))
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        ix.search(q, 40)
        ts.append(time.perf_counter() - t0)
    s0 = ix.stats()
    print(f"sample grid {g:4d}: ms per batch of 256 median {1e3 * np.median(ts):.2f} min {1e3 * min(ts):.2f}; identical to the first config: {same}; proofs (served, failed) {s0}", flush=True)
    ix.close()
    del ix
