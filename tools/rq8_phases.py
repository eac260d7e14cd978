#!/usr/bin/env python3
"""Phase timer of the int8 scan (tools build: libclipx_ablate.so, KNNX_RQ8_TIMER=1): shader cycles per tile in the top-of-tile wait +
barrier, the k-loop and the filter, per wave, for the 4-wave (B <= 128) and the 8-wave (B = 256) forms.   python tools/rq8_phases.py [rows]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ["KNNX_RQ8_TIMER"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import clip_retrieval_amd._lib as L  # noqa: E402

L._LIB_PATH = os.path.join(os.path.dirname(L._LIB_PATH), "libclipx_ablate.so")
from clip_retrieval_amd.knn import Mi355xIndex  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
lib = L.load_library()
raw = C.CDLL(L._LIB_PATH)
ix = Mi355xIndex(768)
ix.synth_fill(rows, 3)
st = torch.cuda.current_stream().cuda_stream
for nq in (64, 128, 256):
    q = torch.nn.functional.normalize(torch.randn(nq, 768, device="cuda"), dim=1)
    D = torch.empty(nq, 40, device="cuda")
    I = torch.empty(nq, 40, device="cuda", dtype=torch.int64)
    for _ in range(2):
        ix.profile(True)
        ix.search_device(q.data_ptr(), nq, 40, D.data_ptr(), I.data_ptr(), st)
        torch.cuda.synchronize()
        ix.profile(False)
        nl, ms = ix.profile_get()
    ph = np.zeros(256 * 8 * 4, dtype=np.int64)
    raw.knnx_dbg_rq8_phases(ph.ctypes.data_as(C.c_void_p), ph.size)
    ph = ph.reshape(256, 8, 4).astype(np.float64)
    nw = 8 if nq > 128 else 4
    tiles = ph[:, :nw, 3].mean()
    per = ph[:, :nw, :3] / np.maximum(ph[:, :nw, 3:4], 1)
    print(f"B={nq}: main pass {ms / max(nl, 1):.3f} ms, {tiles:.0f} tiles per workgroup; cycles per tile (mean over workgroups):")
    for w in range(nw):
        print(f"   wave {w}: wait+barrier {per[:, w, 0].mean():7.0f}   k-loop {per[:, w, 1].mean():7.0f}   filter {per[:, w, 2].mean():7.0f}   sum {per[:, w].sum(axis=1).mean():7.0f}")
ix.close()
