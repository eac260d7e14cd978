"""Multi-GPU layer: one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI).

  * encode half: replicas only.  Output partitions are dealt to ranks with the reference's rule
    (`get_task_list`, clip_retrieval/clip_inference/slurm_worker.py:16-37); no collective.
  * search half: the index is row-sharded (rank g holds rows [g*N/G, (g+1)*N/G), id_base = the first row).
    Every rank scans its shard for the same queries, the per-shard top-k (k x 12 B per query) are exchanged
    with ONE all-gather, and every rank merges P*k -> k with the (score desc, id asc) rule.  This is the
    only exchange step on the path; the reference itself has none (single-process faiss).
The merge runs in lib/libclipx.so (device kernel) for CUDA tensors; host tensors (gloo, CPU tests) are merged with numpy.
The ONE-process-many-GPUs variant of the same sharding (KnnService) is knn.ShardedMi355xIndex / knnx_shards_*.
"""

import ctypes as C

import numpy as np

from ._lib import check, load_library
from .runner import get_task_list  # noqa: F401  (re-export: encode-side task split)


def merge_topk_host(D_parts, I_parts, k):
    """[P, n, k] per-shard results with global ids -> [n, k] by (score desc, id asc), -1 / -FLT_MAX padded.
    numpy, for HOST tensors only: the `gloo` backend of the CPU tests of the exchange logic.  On GPUs (backend "nccl" =
    RCCL) the merge is the device kernel behind knnx_merge_topk_device; the library itself has no CPU arithmetic."""
    D_parts = np.ascontiguousarray(D_parts, dtype=np.float32)
    I_parts = np.ascontiguousarray(I_parts, dtype=np.int64)
    P, n, kk = D_parts.shape
    assert kk == k and I_parts.shape == D_parts.shape
    Dc = np.transpose(D_parts, (1, 0, 2)).reshape(n, P * k)
    Ic = np.transpose(I_parts, (1, 0, 2)).reshape(n, P * k)
    D = np.full((n, k), -np.finfo(np.float32).max, dtype=np.float32)
    I = np.full((n, k), -1, dtype=np.int64)
    for i in range(n):
        ok = Ic[i] >= 0
        order = np.lexsort((Ic[i][ok], -Dc[i][ok].astype(np.float64)))[:k]
        D[i, : order.size] = Dc[i][ok][order]
        I[i, : order.size] = Ic[i][ok][order]
    return D, I


def shard_rows(total_rows, world_size, rank):
    """Contiguous row range of `rank` in a row-sharded index."""
    lo = (total_rows * rank) // world_size
    hi = (total_rows * (rank + 1)) // world_size
    return lo, hi


class ShardedIndex:
    """faiss-shaped `search` over a row-sharded index; collective: every rank must call with the same queries."""

    def __init__(self, local_index, group=None, force_gather=False):
        """`force_gather`: run the exchange step (all-gather + merge) even at world size 1, where a single shard's result is
        already the answer -- so that one GPU can execute the code path N > 1 ranks take (tests; CLIPX_FORCE_GATHER=1 does the
        same from the environment)."""
        import os  # pylint: disable=import-outside-toplevel

        import torch.distributed as dist  # pylint: disable=import-outside-toplevel

        self.local = local_index
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.d = local_index.d
        self.force_gather = bool(force_gather or os.environ.get("CLIPX_FORCE_GATHER") == "1") and dist.is_initialized()

    def _pad_queries(self, x):
        """float32 [n, d] -> contiguous float32 [n, d_padded]: the device entry points read rows of the local index's padded
        width (d rounded up to a multiple of 256; e.g. the 640-dimensional RN50x4 embeddings), zeros in the extra columns."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        dpad = int(getattr(self.local, "_dpad", self.d))
        if x.ndim != 2 or x.shape[1] not in (self.d, dpad):
            raise AssertionError(f"queries must be [n, {self.d}], got {x.shape}")
        if x.shape[1] == dpad:
            return x
        out = np.zeros((x.shape[0], dpad), dtype=np.float32)
        out[:, : self.d] = x
        return out

    @property
    def ntotal(self):
        import torch  # pylint: disable=import-outside-toplevel
        import torch.distributed as dist  # pylint: disable=import-outside-toplevel

        if self.world == 1:
            return self.local.ntotal
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([self.local.ntotal], dtype=torch.int64, device=dev)
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    @staticmethod
    def _record_buffer(n, k, device):
        """One rank's results as ONE buffer of 12-byte records' worth of bytes (a single all-gather of
        B x k x 12 bytes per rank): [n*k int64 ids | n*k float32 scores], padded to a multiple of 8 bytes.  Returns
        (bytes, I view [n, k], D view [n, k])."""
        import torch  # pylint: disable=import-outside-toplevel

        nk = n * k
        size = (nk * 12 + 7) // 8 * 8
        rec = torch.empty(size, dtype=torch.uint8, device=device)
        return rec, rec[: nk * 8].view(torch.int64).view(n, k), rec[nk * 8: nk * 12].view(torch.float32).view(n, k)

    def _gather_records(self, rec, n, k):
        """all-gather of the per-rank record buffers -> (Dg [world, n, k], Ig [world, n, k])."""
        import torch  # pylint: disable=import-outside-toplevel
        import torch.distributed as dist  # pylint: disable=import-outside-toplevel

        nk = n * k
        out = torch.empty(self.world * rec.numel(), dtype=torch.uint8, device=rec.device)
        dist.all_gather_into_tensor(out, rec, group=self.group)
        out = out.view(self.world, rec.numel())
        Ig = out[:, : nk * 8].contiguous().view(torch.int64).view(self.world, n, k)
        Dg = out[:, nk * 8: nk * 12].contiguous().view(torch.float32).view(self.world, n, k)
        return Dg, Ig

    def search(self, x, k):
        """Host queries in, host results out (the clip_back call shape)."""
        import torch  # pylint: disable=import-outside-toplevel
        import torch.distributed as dist  # pylint: disable=import-outside-toplevel

        if self.world == 1 and not self.force_gather:
            return self.local.search(x, k)
        on_gpu = dist.get_backend(self.group) == "nccl"
        if on_gpu and k <= 64 and hasattr(self.local, "search_device"):
            # RCCL: upload the queries once and stay on the device -- local scan into the record buffer, all-gather, merge -- only
            # the merged [n, k] comes back (round 5: the local results used to travel device -> numpy -> device before the gather).
            # A faiss-shaped local index without `search_device` (ShardedMi355xIndex, a test double) takes the host path below.
            dev = torch.device("cuda", torch.cuda.current_device())
            q = torch.from_numpy(self._pad_queries(x)).to(dev, non_blocking=False)
            Do, Io = self.search_device(q, k)
            return Do.cpu().numpy(), Io.cpu().numpy()
        D, I = self.local.search(x, k)
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        n = D.shape[0]
        rec, Iv, Dv = self._record_buffer(n, k, dev)
        Iv.copy_(torch.from_numpy(np.ascontiguousarray(I)))
        Dv.copy_(torch.from_numpy(np.ascontiguousarray(D)))
        Dg, Ig = self._gather_records(rec, n, k)
        if on_gpu:
            Do, Io = self.merge_device(Dg, Ig, k)
            return Do.cpu().numpy(), Io.cpu().numpy()
        return merge_topk_host(Dg.numpy(), Ig.numpy(), k)

    def search_device(self, q_cuda, k):
        """CUDA tensors end to end: local scan -> all_gather (RCCL) -> merge kernel.  Returns (D, I) on the GPU.
        `q_cuda`: float32 [n, d_padded] (see _pad_queries; d itself when d % 256 == 0).

        The scan is launched on a dedicated torch side stream (the C ABI reads a NULL stream as "the index's own
        stream", which torch's collectives would not be ordered against); the caller's current stream waits for it
        before the all-gather reads the local results."""
        import torch  # pylint: disable=import-outside-toplevel
        import torch.distributed as dist  # pylint: disable=import-outside-toplevel

        dpad = int(getattr(self.local, "_dpad", self.d))
        if q_cuda.dim() != 2 or q_cuda.shape[1] != dpad or q_cuda.dtype != torch.float32 or not q_cuda.is_contiguous():
            raise AssertionError(f"search_device expects contiguous float32 [n, {dpad}] queries (padded width), got {tuple(q_cuda.shape)} {q_cuda.dtype}")
        n = q_cuda.shape[0]
        rec, I, D = self._record_buffer(n, k, q_cuda.device)  # the local scan writes straight into the record buffer
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=q_cuda.device)
        cur = torch.cuda.current_stream(q_cuda.device)
        self._side.wait_stream(cur)  # q_cuda, D, I were produced / allocated on the caller's stream
        self.local.search_device(q_cuda.data_ptr(), n, k, D.data_ptr(), I.data_ptr(), self._side.cuda_stream)
        cur.wait_stream(self._side)
        if self.world == 1 and not self.force_gather:
            return D, I
        Dg, Ig = self._gather_records(rec, n, k)
        return self.merge_device(Dg, Ig, k)

    @staticmethod
    def merge_device(Dg, Ig, k):
        import torch  # pylint: disable=import-outside-toplevel

        lib = load_library()
        P, n, kk = Dg.shape
        assert kk == k
        Do = torch.empty((n, k), dtype=torch.float32, device=Dg.device)
        Io = torch.empty((n, k), dtype=torch.int64, device=Dg.device)
        st = torch.cuda.current_stream(Dg.device).cuda_stream
        check(lib, lib.knnx_merge_topk_device(Dg.device.index or 0, C.c_void_p(Dg.data_ptr()), C.c_void_p(Ig.data_ptr()),
                                              P, n, k, C.c_void_p(Do.data_ptr()), C.c_void_p(Io.data_ptr()),
                                              C.c_void_p(st)), "knnx")
        return Do, Io
