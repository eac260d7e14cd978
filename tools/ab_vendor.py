#!/usr/bin/env python3
"""Calibration, not product: this library's bf16 GEMM against the vendor's (hipBLASLt behind torch.nn.functional.linear) on
the SAME box, the SAME random operands (uniform [-1, 1) activations, 0.05 x that for the weights, f32 bias), the four ViT-L/14
bs = 256 shapes, alternating launch by launch.  Two regimes per shape:
  isolated   one launch, host sync, next launch (what tools/gemm_bench and tools/calib_blas.py time): median of REPS
  sustained  BURST launches back to back per arm (the encoder's regime: the part stays at its power limit): ms per launch
hipBLASLt is never linked by libclipx.so; this only says how far from the practical ceiling the hand-written kernel is."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clip_retrieval_amd import load_library  # noqa: E402

lib = load_library()
P = lambda t: C.c_void_p(t.data_ptr())
REPS = int(os.environ.get("AB_REPS", "20"))
BURST = int(os.environ.get("AB_BURST", "40"))
M = int(os.environ.get("AB_M", "65536"))
shapes = [("QKV", M, 3072, 1024), ("fc1", M, 4096, 1024), ("out-proj", M, 1024, 1024), ("fc2", M, 1024, 4096)]
only = os.environ.get("AB_ONLY")
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, m, N, K in shapes:
    if only and name not in only.split(","):
        continue
    A = (torch.rand(m, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).to(torch.bfloat16)
    b = torch.rand(N, device="cuda") * 2 - 1
    b16 = b.to(torch.bfloat16)
    out = torch.empty(m, N, device="cuda", dtype=torch.bfloat16)

    def ours():
        rc = lib.clipx_gemm_bf16_device(0, P(A), P(W), P(b), P(out), m, N, K, 0, C.c_void_p(st))
        assert rc == 0, lib.clipx_last_error()

    def vendor():
        torch.nn.functional.linear(A, W, b16, ) if os.environ.get("AB_VENDOR_BIAS") else torch.nn.functional.linear(A, W)

    for _ in range(3):
        ours(); vendor()
    torch.cuda.synchronize()
    ref = torch.nn.functional.linear(A, W).float() + b
    ours()
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    iso = {"ours": [], "vendor": []}
    for _ in range(REPS):
        iso["ours"].append(timed(ours))
        iso["vendor"].append(timed(vendor))
    sus = {"ours": [], "vendor": []}
    for _ in range(3):
        sus["ours"].append(timed(ours, BURST))
        sus["vendor"].append(timed(vendor, BURST))
    fl = 2.0 * m * N * K
    tf = lambda ms: fl / ms / 1e9
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{name:9s} {m}x{N}x{K}  max |ours - vendor| {err:.3g}", flush=True)
    for arm in ("ours", "vendor"):
        print(f"   {arm:7s} isolated median {med(iso[arm]):.4f} ms {tf(med(iso[arm])):7.1f} TF (min {tf(min(iso[arm])):7.1f} TF)   "
              f"sustained x{BURST}: {med(sus[arm]):.4f} ms {tf(med(sus[arm])):7.1f} TF", flush=True)
