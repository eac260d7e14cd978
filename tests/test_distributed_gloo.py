"""world_size-2 gloo tests of the N>1 search path on CPU: shard split, all-gather, (score desc, id asc) merge.
The per-shard scan is the GPU kernel in production; here a stand-in local index (the numpy oracle, which only
tests may use) supplies per-shard results so that the collective + merge logic is what is exercised."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _LocalOracleShard:
    """faiss-shaped local index over rows [lo, hi) returning GLOBAL ids, like Mi355xIndex with id_base=lo."""

    def __init__(self, rows, lo):
        from oracle.knn_oracle import FlatIPOracle

        self.o = FlatIPOracle(rows.shape[1])
        self.o.add(rows)
        self.lo, self.d = lo, rows.shape[1]

    @property
    def ntotal(self):
        return self.o.ntotal

    def search(self, x, k):
        D, I = self.o.search(x, k)
        return D, np.where(I >= 0, I + self.lo, -1)


def _worker(rank, world, port, n_rows, k, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clip_retrieval_amd.distributed import ShardedIndex, get_task_list, shard_rows

        rng = np.random.default_rng(123)
        X = rng.standard_normal((n_rows, 32)).astype(np.float16)
        X[n_rows - 1] = X[0]  # an exact tie that straddles the two shards
        q = rng.standard_normal((5, 32)).astype(np.float32)
        q[0] = X[0].astype(np.float32)
        lo, hi = shard_rows(n_rows, world, rank)
        idx = ShardedIndex(_LocalOracleShard(X[lo:hi], lo))
        assert idx.ntotal == n_rows
        D, I = idx.search(q, k)
        np.save(os.path.join(out_dir, f"D{rank}.npy"), D)
        np.save(os.path.join(out_dir, f"I{rank}.npy"), I)
        # encode-side split: every partition goes to exactly one rank
        mine = get_task_list(11, world, rank)
        t = torch.zeros(11, dtype=torch.int64)
        t[mine] = 1
        dist.all_reduce(t)
        assert t.tolist() == [1] * 11
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows,k", [(101, 7), (9, 12)])
def test_sharded_search_matches_single_index(tmp_path, n_rows, k):
    from oracle.knn_oracle import FlatIPOracle

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_rows, k, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    X = rng.standard_normal((n_rows, 32)).astype(np.float16)
    X[n_rows - 1] = X[0]
    q = rng.standard_normal((5, 32)).astype(np.float32)
    q[0] = X[0].astype(np.float32)
    full = FlatIPOracle(32)
    full.add(X)
    Do, Io = full.search(q, k)
    for r in range(world):
        D, I = np.load(tmp_path / f"D{r}.npy"), np.load(tmp_path / f"I{r}.npy")
        assert np.array_equal(I, Io), (I, Io)
        assert np.array_equal(D, Do)
    # the straddling tie is returned in id order
    pos = {int(v): j for j, v in enumerate(Io[0])}
    if n_rows - 1 in pos:
        assert pos[0] < pos[n_rows - 1]


def test_shard_rows_partition():
    from clip_retrieval_amd.distributed import shard_rows

    for total in (0, 1, 7, 1000, 10**9):
        for world in (1, 2, 8):
            spans = [shard_rows(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_device_fast_path_pads_queries_to_the_local_index_width():
    """the RCCL fast path hands the queries to `local.search_device`, which reads rows of the PADDED width (d rounded up to
    a multiple of 256: 640-dimensional RN50x4 embeddings -> 768); a local index without `search_device` must not take that path."""
    from clip_retrieval_amd.distributed import ShardedIndex

    class _Padded:
        d, _dpad, ntotal = 640, 768, 0

        def search_device(self, *a):  # pragma: no cover - never called on the CPU
            raise AssertionError

    sh = ShardedIndex(_Padded())
    q = np.random.default_rng(0).standard_normal((3, 640)).astype(np.float32)
    p = sh._pad_queries(q)
    assert p.shape == (3, 768) and p.dtype == np.float32 and p.flags.c_contiguous
    assert np.array_equal(p[:, :640], q) and not p[:, 640:].any()
    assert sh._pad_queries(p) is p  # already padded: passed through
    with pytest.raises(AssertionError):
        sh._pad_queries(q[:, :100])
    with pytest.raises(AssertionError):  # search_device refuses an unpadded tensor instead of reading past its rows
        sh.search_device(torch.zeros(3, 640), 5)
    assert not hasattr(_LocalOracleShard, "search_device")  # the gloo stand-in goes through local.search + gather
