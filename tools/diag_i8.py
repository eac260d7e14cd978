"""Diagnostic: the int8 first stage at scale -- planted neighbours, per query, for several index sizes and batch sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_retrieval_amd.knn import Mi355xIndex
from oracle.knn_oracle import planted_queries

d, seed = 768, 3
for n in [int(x) for x in sys.argv[1:]] or [10_000_000, 100_000_000]:
    ix = Mi355xIndex(d)
    ix.synth_fill(n, seed)
    rng = np.random.default_rng(1)
    planted = np.sort(rng.choice(n, 256, replace=False))
    q = planted_queries(planted, d, seed)
    for nq in (32, 64, 128, 256, 256):
        s0 = ix.stats(); i0 = ix.i8_served()
        D, I = ix.search(q[:nq], 40)
        s1 = ix.stats()
        bad = np.nonzero(I[:, 0] != planted[:nq])[0]
        print(f"n={n} nq={nq}: i8 served {ix.i8_served() - i0}, proofs failed {s1[1] - s0[1]}, wrong top-1: {len(bad)} {bad[:12].tolist()}", flush=True)
        for b in bad[:3]:
            print("    query", b, "planted", planted[b], "got", I[b, :3].tolist(), D[b, :3].tolist())
    ix.close()
