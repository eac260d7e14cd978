"""Diagnostic: knn_assign_kernel against torch argmax at several (d, nlist), then per-iteration list sizes of the device
k-means on the mixture corpus (round 3: config 5's lists came out badly unbalanced)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from clip_retrieval_amd.knn import IvfBuilder, synth_rows_device, train_ivf_centroids_device

dev = "cuda:0"
for d in (1024, 768, 512):
    for nlist in (33, 64, 96, 128, 1024, 16384):
        n = 20000
        g = torch.Generator(device=dev).manual_seed(nlist + d)
        x = torch.randn(n, d, generator=g, device=dev)
        x = (x / x.norm(dim=1, keepdim=True)).half()
        c = torch.randn(nlist, d, generator=g, device=dev)
        c = (c / c.norm(dim=1, keepdim=True)).half()
        b = IvfBuilder(d, nlist)
        b.set_centroids(c.cpu().numpy())
        out = torch.empty(n, dtype=torch.int32, device=dev)
        b.assign_device(x.data_ptr(), n, out.data_ptr())
        s = x.float() @ c.float().T
        want = s.argmax(1)
        got = out.long()
        bad = (got != want)
        # near ties do not count
        gap = s.gather(1, want[:, None])[:, 0] - s.gather(1, got.clamp(0, nlist - 1)[:, None])[:, 0]
        real_bad = int((bad & (gap > 1e-4)).sum())
        print(f"assign d={d} nlist={nlist}: {int(bad.sum())} differ, {real_bad} beyond near-ties, got range [{int(got.min())}, {int(got.max())}]", flush=True)
        b.close()

for d in (1024, 768):
    n, nlist, ncl = 1 << 20, 1024, 128
    buf = torch.empty((n, d), dtype=torch.float16, device=dev)
    synth_rows_device(buf.data_ptr(), 0, n, d, 5, kind=1, n_clusters=ncl)
    ns = nlist * 64
    sample = buf[:: n // ns][:ns].contiguous()
    b = IvfBuilder(d, nlist)
    rng = np.random.default_rng(1)
    b.set_sample_device(sample.data_ptr(), ns)
    b.seed_from_sample(np.arange(nlist), np.sort(rng.choice(ns, nlist, replace=False)))
    for it in range(8):
        sizes = b.lloyd()
        cent = torch.from_numpy(b.centroids().astype(np.float32))
        print(f"kmeans d={d} iter {it}: empty {int((sizes == 0).sum())} sizes min/med/max {sizes.min()} {int(np.median(sizes))} {sizes.max()} "
              f"centroid norm median {float(cent.norm(dim=1).median()):.3f}", flush=True)
        e = np.flatnonzero(sizes == 0)
        if e.size:
            b.seed_from_sample(e, rng.choice(ns, e.size, replace=False))
    lists = torch.empty(n, dtype=torch.int32, device=dev)
    b.assign_device(buf.data_ptr(), n, lists.data_ptr())
    sz = np.bincount(lists.cpu().numpy(), minlength=nlist)
    want = (buf.float() @ torch.from_numpy(b.centroids().astype(np.float32)).to(dev).T).argmax(1)
    print(f"kmeans d={d} full assignment: sizes min/med/max {sz.min()} {int(np.median(sz))} {sz.max()}; vs torch argmax: {int((want != lists.long()).sum())} differ", flush=True)
    b.close()
