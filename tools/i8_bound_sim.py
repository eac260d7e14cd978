"""numpy simulation behind DESIGN 4.3: how wide the admission band of the int8 first stage is, in sigmas of the score distribution,
for isotropic unit vectors and for embeddings with a few dominant columns, with one and with two int8 planes per query -- and how many
rows per query that admits at 10^8 rows for a threshold at z = 4 (about rank 3 000).  CPU only:  python tools/i8_bound_sim.py"""
import os
import sys

import numpy as np
from scipy.stats import norm

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.knn_oracle import Int8FirstStage  # noqa: E402  (test infrastructure: this tool is a simulation, not product code)

rng = np.random.default_rng(0)
d, n = 768, 40000


def run(label, rogue_scale, nrogue):
    x = rng.standard_normal((n, d)).astype(np.float32)
    if nrogue:
        x[:, :nrogue] = rogue_scale * x[:, :nrogue] + 0.5 * rogue_scale  # large spread + a common offset, as CLIP embeddings have
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    st = Int8FirstStage(x.astype(np.float16))
    q = x[:256] + 0.1 * rng.standard_normal((256, d)).astype(np.float32) / np.sqrt(d)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    sig = float((q @ st.x[1000:].T).std())
    ratio = float(st.c.max() / np.median(st.c))
    out = [f"{label}: largest / median column scale {ratio:.1f}, score sigma {sig:.4f}"]
    for planes in (1, 2):
        s, pl, eps8 = st.quantise_queries(q, planes)
        exact = q.astype(np.float64) @ st.x.astype(np.float64).T
        err = np.abs(exact - s[:, None].astype(np.float64) * st.integer_scores(pl))
        band = float(eps8.mean()) / sig
        out.append(f"    {planes} plane(s): eps8 {eps8.mean():.4f} = {band:.2f} sigma (largest actual error {err.max():.4f}); "
                   f"rows admitted at 1e8 rows, threshold z = 4.0: {int(norm.sf(4.0 - band) * 1e8):,}")
    print("\n".join(out), flush=True)


run("isotropic", 1.0, 0)
run("2 dominant columns (x5)", 5.0, 2)
run("2 dominant columns (x15)", 15.0, 2)
