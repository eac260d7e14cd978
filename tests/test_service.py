"""Request hot path of clip_back restated in clip-retrieval_amd/service.py: metadata join (f3), connected components and
post-filter logic on the CPU; the GPU dedup / knn_search parity tests are marked gpu."""
import numpy as np
import pytest


def _reference_arrow_get(table, ids, cols):
    """The reference's ArrowMetadataProvider.get, verbatim semantics (clip_back.py:608-615)."""
    import pyarrow as pa

    cols = table.schema.names if cols is None else list(set(table.schema.names) & set(cols))
    t = pa.concat_tables([table[i: i + 1] for i in ids])
    return t.select(cols).to_pandas().to_dict("records")


def test_arrow_metadata_provider_batched_take_equals_reference_slicing(tmp_path):
    import pyarrow as pa

    from clip_retrieval_amd.service import ArrowMetadataProvider

    rng = np.random.default_rng(0)
    for f in range(3):  # three IPC files, concatenated in sorted order like the reference
        n = 500 + f
        t = pa.table({"url": [f"http://x/{f}/{i}" for i in range(n)], "caption": [f"cap {f} {i}" for i in range(n)],
                      "width": rng.integers(0, 4000, n), "similarity": rng.random(n).astype(np.float32)})
        with pa.OSFile(str(tmp_path / f"{f}.arrow"), "wb") as sink:
            with pa.ipc.new_file(sink, t.schema) as w:
                w.write_table(t, max_chunksize=100)
    p = ArrowMetadataProvider(str(tmp_path))
    assert p.table.num_rows == 1503
    for ids, cols in [([5, 700, 3, 1502, 3], ["url", "caption"]), ([0], None), (list(range(1000, 1040)), ["url", "nope"]), ([], ["url"])]:
        got = p.get(ids, cols)
        want = _reference_arrow_get(p.table, ids, cols) if ids else []
        assert got == want


def _reference_non_uniques(neighbors):
    """clip_back.py:270-309 restated as the checker: depth-first groups over the range-search links, every group keeps the
    first node the walk reaches (nodes are visited in ascending order), the rest are non-unique."""
    seen, out = set(), set()
    for start in neighbors:
        if start in seen:
            continue
        todo, group = {start}, []
        while todo:
            node = todo.pop()
            seen.add(node)
            todo |= set(neighbors.get(node, ())) - seen
            group.append(node)
        out |= set(group[1:])
    return out


def test_dedup_groups_from_range_search_links():
    """service.non_uniques_from_links (CSR links -> scipy connected components, a group keeps its smallest index) against
    the reference's DFS on hand-made and random link structures."""
    from clip_retrieval_amd.service import KnnHotPath

    nb = {0: [0, 2], 1: [1], 2: [2, 0, 4], 3: [3], 4: [4, 2], 5: [5, 6], 6: [6, 5]}
    lims = np.cumsum([0] + [len(nb[i]) for i in range(7)])
    nbr = np.concatenate([nb[i] for i in range(7)])
    assert set(KnnHotPath.non_uniques_from_links(lims, nbr, 7)) == _reference_non_uniques(nb) == {2, 4, 6}
    rng = np.random.default_rng(0)
    for n in (1, 5, 60, 300):
        a = rng.random((n, n)) < 1.5 / n
        a = a | a.T | np.eye(n, dtype=bool)
        nb = {i: np.flatnonzero(a[i]).tolist() for i in range(n)}
        lims = np.cumsum([0] + [len(nb[i]) for i in range(n)])
        nbr = np.concatenate([nb[i] for i in range(n)]) if n else np.zeros(0, int)
        assert set(KnnHotPath.non_uniques_from_links(lims, nbr, n)) == _reference_non_uniques(nb)


def test_dedup_groups_from_pair_links_and_the_request_tail_without_a_gpu():
    """Round 4: (1) `non_uniques_from_pairs` (the links knnx_search_dedup reports: pairs i < j) against the reference's DFS on random
    link structures; (2) `KnnHotPath.knn_search`'s result handling on a FAKE index (no GPU): the plain-Python fast path (nothing
    dropped, ids distinct), the -1 truncation, duplicate ids reported once at their best rank, the fused-dedup drop set, and the
    safety / violence filters must give what the reference's loop gives (clip_back.py:371-399 restated below)."""
    from types import SimpleNamespace

    from clip_retrieval_amd.service import KnnHotPath, normalized

    rng = np.random.default_rng(1)
    for n in (2, 7, 40, 64):
        for dens in (0.0, 0.5 / n, 2.0 / n):
            a = np.triu(rng.random((n, n)) < dens, 1)
            pairs = np.argwhere(a).astype(np.int32)
            nb = {i: sorted(set([i] + np.flatnonzero(a[i] | a[:, i]).tolist())) for i in range(n)}
            assert set(KnnHotPath.non_uniques_from_pairs(pairs, n)) == _reference_non_uniques(nb), (n, dens)

    def reference_tail(D, I, to_remove_local):
        results = I[0]
        nb = np.where(results == -1)[0]
        n = nb[0] if len(nb) else len(results)
        ri, rd = results[:n], D[0][:n]
        removed = set(ri[j] for j in to_remove_local if j < n)
        out_i, out_d = [], []
        for ind, dist in zip(ri, rd):
            if ind not in removed:
                removed.add(ind)
                out_i.append(ind)
                out_d.append(dist)
        return out_d, out_i

    class FakeIndex:
        d = 8

        def __init__(self, D, I, links):
            self.D, self.I, self.links = D, I, links
            self.R = rng.standard_normal((1, I.shape[1], 8)).astype(np.float32)
            self.calls = []

        def search(self, q, k):
            self.calls.append("search")
            return self.D.copy(), self.I.copy()

        def search_and_reconstruct(self, q, k):
            self.calls.append("search_and_reconstruct")
            return self.D.copy(), self.I.copy(), self.R.copy()

        def search_dedup(self, q, k, thr, want_r=False):
            self.calls.append("search_dedup" + ("+R" if want_r else ""))
            return self.D.copy(), self.I.copy(), (self.R.copy() if want_r else None), self.links.copy()

    class Safety:  # flags result rank 1
        def predict(self, emb, batch_size=None):
            y = np.zeros((emb.shape[0], 1), np.float32)
            y[1] = 1.0
            return y

    hp = KnnHotPath.__new__(KnnHotPath)
    import threading

    hp._nprobe_lock = threading.Lock()  # pylint: disable=protected-access
    q = np.zeros((1, 8), np.float32)
    k = 12
    D = np.sort(rng.random((1, k)).astype(np.float32))[:, ::-1].copy()
    cases = {
        "distinct": np.arange(100, 100 + k, dtype=np.int64)[None],
        "short": np.r_[np.arange(5, 12), -np.ones(5)].astype(np.int64)[None],
        "repeated ids": np.asarray([[4, 9, 4, 7, 9, 1, 2, 3, 5, 6, 8, 4]], dtype=np.int64),
    }
    for name, I in cases.items():
        for links in (np.zeros((0, 2), np.int32), np.asarray([[0, 3], [3, 6], [2, 5]], np.int32)):
            ix = FakeIndex(D, I, links)
            res = SimpleNamespace(image_index=ix, text_index=ix, metadata_is_ordered_by_ivf=False, safety_model=None, violence_detector=None)
            n = int(np.argmax(I[0] == -1)) if (I[0] == -1).any() else k
            d0, i0 = hp.knn_search(q, "image", k, res, False, False, False)
            assert ix.calls == ["search"], "no filter reads the vectors: they must not be fetched"
            rd, ri = reference_tail(D, I, [])
            assert [int(v) for v in i0] == [int(v) for v in ri] and np.array_equal(np.asarray(d0), np.asarray(rd)), name
            ix.calls.clear()
            d1, i1 = hp.knn_search(q, "image", k, res, True, False, False)
            assert ix.calls == ["search_dedup"]
            drop = KnnHotPath.non_uniques_from_pairs(links[(links[:, 0] < n) & (links[:, 1] < n)], n)
            rd, ri = reference_tail(D, I, drop)
            assert [int(v) for v in i1] == [int(v) for v in ri] and np.array_equal(np.asarray(d1), np.asarray(rd)), (name, len(links))
            ix.calls.clear()
            res.safety_model = Safety()
            d2, i2 = hp.knn_search(q, "image", k, res, True, True, False)
            assert ix.calls == ["search_dedup+R"]
            rd, ri = reference_tail(D, I, sorted(set(drop) | ({1} if n > 1 else set())))
            assert [int(v) for v in i2] == [int(v) for v in ri], (name, "safety")
            assert all(hasattr(v, "item") for v in i2) and all(hasattr(v, "item") for v in d2)  # numpy scalars, like the reference's lists


def test_map_to_metadata_and_embedding_query(tmp_path):
    """map_to_metadata (clip_back.py:401-417): metadata for the first num_images ids only, id / similarity always, bytes
    decoded; compute_query's embedding branch with the aesthetic shift (clip_back.py:247-255)."""
    from types import SimpleNamespace

    from clip_retrieval_amd.service import KnnHotPath

    class Provider:
        def get(self, ids, cols=None):
            return [{"url": f"u{int(i)}".encode(), "caption": f"c{int(i)}", "skip": 1} if cols is None else
                    {k: v for k, v in {"url": f"u{int(i)}".encode(), "caption": f"c{int(i)}"}.items() if k in cols} for i in ids]

    ids, dist = [np.int64(7), np.int64(3), np.int64(9)], [np.float32(0.9), np.float32(0.8), np.float32(0.7)]
    got = KnnHotPath.map_to_metadata(ids, dist, 2, Provider(), ["url", "caption"])
    assert got == [{"url": "u7", "caption": "c7", "id": 7, "similarity": float(np.float32(0.9))},
                   {"url": "u3", "caption": "c3", "id": 3, "similarity": float(np.float32(0.8))},
                   {"id": 9, "similarity": float(np.float32(0.7))}]
    hp = KnnHotPath.__new__(KnnHotPath)  # no GPU needed for the embedding branch
    aest = np.zeros((10, 4), np.float32)
    aest[9] = [0, 1, 0, 0]
    res = SimpleNamespace(aesthetic_embeddings=aest)
    q = hp.compute_query(res, None, None, None, [1.0, 0.0, 0.0, 0.0], False, 9, 0.5)
    want = np.array([[1.0, 0.5, 0, 0]], np.float32)
    assert q.shape == (1, 4) and q.dtype == np.float32 and np.allclose(q, want / np.linalg.norm(want))
    assert np.array_equal(hp.compute_query(SimpleNamespace(aesthetic_embeddings=None), None, None, None, [0.0, 2.0], False, None, None),
                          np.array([[0.0, 2.0]], np.float32))


@pytest.mark.gpu
def test_gpu_dedup_and_knn_search_match_the_reference_logic():
    """get_non_uniques on the GPU (range scan over the <= k result vectors) against a numpy restatement of
    clip_back.py:290-309, then knn_search end to end (ordered unique ids, -1 truncation, dedup applied) against the same
    logic over the flat oracle."""
    from types import SimpleNamespace

    from clip_retrieval_amd.knn import Mi355xIndex
    from clip_retrieval_amd.service import KnnHotPath, normalized
    from oracle.knn_oracle import FlatIPOracle

    rng = np.random.default_rng(3)
    d, n = 512, 4000
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x[100:110] = x[7] + 0.01 * rng.standard_normal((10, d)).astype(np.float32)  # a cluster of near-duplicates of row 7
    x[500] = x[499]
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x16 = x.astype(np.float16)
    hp = KnnHotPath()

    def ref_non_uniques(emb, thr=0.94):
        s = emb @ emb.T
        return _reference_non_uniques({i: [int(j) for j in np.flatnonzero(s[i] > thr)] for i in range(emb.shape[0])})

    R = normalized(x16[np.r_[0:20, 100:110, 499:501, 7]].astype(np.float32))
    assert set(hp.get_non_uniques(R)) == ref_non_uniques(R)
    assert hp.get_non_uniques(R[:1]) == [] and hp.get_non_uniques(np.zeros((0, d), np.float32)) == []
    # the front end's k = 3000: 94 groups of 32 result vectors, launched back to back (knnx_range_search_once, batched)
    big = normalized(x16[:3000].astype(np.float32))
    assert set(hp.get_non_uniques(big)) == ref_non_uniques(big)

    ix, o = Mi355xIndex(d), FlatIPOracle(d)
    ix.add(x16)
    o.add(x16)
    res = SimpleNamespace(image_index=ix, text_index=ix, metadata_is_ordered_by_ivf=False, safety_model=None, violence_detector=None)
    q = x[7:8].copy()
    for dedup in (False, True):
        dist, ind = hp.knn_search(q, "image", 40, res, dedup, False, False)
        Do, Io, Ro = o.search_and_reconstruct(q, 40)
        keep = [i for i in range(40) if not dedup or i not in ref_non_uniques(normalized(Ro[0]))]
        assert [int(v) for v in ind] == [int(Io[0][i]) for i in keep]
        assert np.allclose(dist, Do[0][keep], atol=1e-5)
    dist, ind = hp.knn_search(q, "image", 5000, res, False, False, False)  # k > ntotal: truncated at the first -1
    assert len(ind) == n and len(set(int(v) for v in ind)) == n
    ix.close()
