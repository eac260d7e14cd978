"""f32 accumulators of the tail against the 128 x 128 kernel: epi 3 (out f32 += acc + bias) on a zeroed output, bf16 operands."""
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
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.randn(M, K, generator=g, device="cuda") * 1.0).to(torch.bfloat16)
W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
bias = torch.zeros(N, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
outs = {}
for v in (1, 3):
    os.environ["CLIPX_GEMM_VARIANT"] = str(v)
    y = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    check(lib, lib.clipx_gemm_bf16_ex_device(0, P(A), P(W), P(bias), P(y), M, N, K, 3, None, None, None), "clipx")
    torch.cuda.synchronize()
    outs[v] = y
d = (outs[1].view(torch.int32) != outs[3].view(torch.int32))
print("differing f32 outputs:", int(d.sum()), "of", d.numel(), "; in the last 256 rows:", int(d[-256:].sum()))
t = d[-256:]
print("tail rows % 32 with differences:", torch.bincount(t.nonzero()[:, 0] % 32, minlength=32).tolist())
print("tail cols % 32 with differences:", torch.bincount(t.nonzero()[:, 1] % 32, minlength=32).tolist())
rel = ((outs[1] - outs[3]).abs() / outs[1].abs().clamp_min(1e-6))[-256:]
print("max rel diff", float(rel.max()))
ref = (A[-256:].double() @ W.double().T)
for v in (1, 3):
    print("variant", v, "max err vs f64 on the tail rows", float((outs[v][-256:].double() - ref).abs().max()))
