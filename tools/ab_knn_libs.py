#!/usr/bin/env python3
"""A/B of two builds of the library on the flat scans (one process per build, same box): python tools/ab_knn_libs.py <lib.so> [rows]
Prints per batch size the whole-call time and the main scan kernel's, int8 first stage on and (KNNX_I8=0) off."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import clip_retrieval_amd._lib as L  # noqa: E402

L._LIB_PATH = os.path.abspath(sys.argv[1])
from clip_retrieval_amd.knn import Mi355xIndex  # noqa: E402

rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
st = torch.cuda.current_stream().cuda_stream
for i8 in ("1", "0"):
    os.environ["KNNX_I8"] = i8
    ix = Mi355xIndex(768)
    ix.synth_fill(rows, 3)
    for nq in ((64, 128, 256) if i8 == "1" else (256,)):
        q = torch.nn.functional.normalize(torch.randn(nq, 768, device="cuda", generator=torch.Generator(device="cuda").manual_seed(nq)), dim=1)
        D = torch.empty(nq, 40, device="cuda")
        I = torch.empty(nq, 40, device="cuda", dtype=torch.int64)
        call = lambda: ix.search_device(q.data_ptr(), nq, 40, D.data_ptr(), I.data_ptr(), st)
        call(); torch.cuda.synchronize()
        ix.profile(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record(); torch.cuda.synchronize()
        ix.profile(False)
        nl, ms = ix.profile_get()
        print(f"{os.path.basename(sys.argv[1]):22s} i8={i8} B={nq:3d}: {e0.elapsed_time(e1) / 5:7.3f} ms per batch, main pass {ms / max(nl, 1):7.3f} ms, failures {ix.stats()[1]}", flush=True)
    ix.close()
