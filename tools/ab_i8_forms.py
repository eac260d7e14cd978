#!/usr/bin/env python3
"""The forms of the int8 first stage on the corpus with dominant columns (knnx_synth_rows_device kind 2), same box, same rows:
dominant columns as 14-bit digits (round 5 default), two query planes (KNNX_I8_DOM=0, round 4), one plain plane (KNNX_I8_PLANES=1).
python tools/ab_i8_forms.py [rows]  ->  per form and batch size: ms per batch, main pass ms, hits admitted, fallbacks."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clip_retrieval_amd.knn import Mi355xIndex, synth_rows_device  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
d, k = 768, 40
X = torch.empty((rows, d), dtype=torch.float16, device="cuda")
synth_rows_device(X.data_ptr(), 0, rows, d, 3, kind=2, device=0)
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(7)
pick = torch.from_numpy(np.sort(rng.choice(rows, 256, replace=False))).cuda()
q = X[pick].float() + 0.3 * torch.randn(256, d, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) / d ** 0.5
q = torch.nn.functional.normalize(q, dim=1).contiguous()
ref = {}
for name, env in (("dominant digits", {}), ("two planes", {"KNNX_I8_DOM": "0"}), ("one plain plane", {"KNNX_I8_PLANES": "1"})):
    for kk, v in env.items():
        os.environ[kk] = v
    ix = Mi355xIndex(d)
    ix.attach_device_rows(X.data_ptr(), rows)
    for nq in (1, 64, 128, 256):
        D = torch.empty(nq, k, device="cuda")
        I = torch.empty(nq, k, device="cuda", dtype=torch.int64)
        call = lambda: ix.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st)
        call(); torch.cuda.synchronize()
        if nq not in ref:
            ref[nq] = (D.clone(), I.clone())
        same = bool((I == ref[nq][1]).all()) and bool((D == ref[nq][0]).all())
        s0, i0 = ix.stats(), ix.i8_served()
        ix.profile(True)
        t0 = time.perf_counter()
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 5 * 1e3
        ix.profile(False)
        nl, ms = ix.profile_get()
        s1 = ix.stats()
        print(f"{name:16s} planes={ix.i8_planes()} dominant={ix.i8_dominant()} B={nq:3d}: {per:7.3f} ms per batch, "
              f"main pass {ms / max(nl, 1):7.3f} ms x {nl / 5:.1f}, int8-served {ix.i8_served() - i0 == 5 * nq}, fallbacks {s1[1] - s0[1]}, "
              f"top-1 planted {bool((I[:, 0] == pick[:nq]).all())}, equal to the first form {same}", flush=True)
    ix.close()
    for kk in env:
        del os.environ[kk]
