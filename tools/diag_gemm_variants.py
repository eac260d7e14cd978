"""Where do the GEMM kernels differ?  The same fp16-operand GEMM (QKV-shaped, epi 7) through CLIPX_GEMM_VARIANT 0 / 1 / 3; prints the
count and position pattern of differing outputs pairwise and the error of each against an f64 product."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clip_retrieval_amd  # noqa: E402
from clip_retrieval_amd._lib import check  # noqa: E402

lib = clip_retrieval_amd.load_library()
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (65792, 3072, 1024))]
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 7
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.randn(M, K, generator=g, device="cuda") * 1.0).to(torch.float16)
W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(torch.float16)
bias = torch.randn(N, generator=g, device="cuda")
rs = torch.rand(M, generator=g, device="cuda") * 0.3 + 0.05
if os.environ.get("DIAG_UNIT"):
    rs.fill_(1.0)
    bias.zero_()
P = lambda t: C.c_void_p(t.data_ptr())
outs = {}
for v in (0, 1, 3):
    os.environ["CLIPX_GEMM_VARIANT"] = str(v)
    y = torch.empty(M, N, device="cuda", dtype=torch.float16 if epi == 7 else torch.bfloat16)
    check(lib, lib.clipx_gemm_f16_device(0, P(A), P(W), P(bias), P(y), M, N, K, epi, P(rs), None), "clipx")
    torch.cuda.synchronize()
    outs[v] = y
for a, b in ((0, 1), (0, 3), (1, 3)):
    d = (outs[a].view(torch.int16) != outs[b].view(torch.int16)).nonzero()
    print(f"variant {a} vs {b}: {d.shape[0]} differing outputs", d[:6].tolist())
    if d.shape[0]:
        r, c = d[:, 0], d[:, 1]
        print("   rows min/max", int(r.min()), int(r.max()), " rows % 32 hist", torch.bincount(r % 32, minlength=32).tolist())
        print("   cols % 32 hist", torch.bincount(c % 32, minlength=32).tolist())
        print("   abs diff max", float((outs[a].float() - outs[b].float())[r, c].abs().max()))
rows = torch.cat([torch.arange(0, 64), torch.arange(M - 256, M)]).cuda()
ref = (A[rows].double() @ W.double().T) * rs[rows, None].double() + bias.double()
for v in (0, 1, 3):
    e = (outs[v][rows].double() - ref).abs()
    print(f"variant {v}: max err vs f64 on the first 64 and last 256 rows {float(e.max()):.3e}, mean {float(e.mean()):.3e}")
