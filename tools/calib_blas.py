#!/usr/bin/env python3
"""Calibration, not product: what the vendor GEMM (hipBLASLt behind torch.nn.functional.linear) reaches on this box for the four
ViT-L/14 bs=256 shapes, same operand layout (A [M,K], W [N,K], bf16, f32 accumulate), random normal data, no epilogue.
The library is never linked by libclipx.so; this only tells how far from the practical ceiling the hand-written kernel is."""
import torch

shapes = [("QKV", 65536, 3072, 1024), ("fc1", 65536, 4096, 1024), ("out-proj", 65536, 1024, 1024), ("fc2", 65536, 1024, 4096)]
torch.manual_seed(0)
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    W = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.03
    for _ in range(3):
        torch.nn.functional.linear(A, W)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.nn.functional.linear(A, W)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    fl = 2.0 * M * N * K
    print(f"vendor GEMM {name:9s} {M}x{N}x{K}: median {ts[10]:.4f} ms {fl / ts[10] / 1e9:7.1f} TF   (min {ts[0]:.4f} ms {fl / ts[0] / 1e9:7.1f} TF)", flush=True)
