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


def test_connected_components_and_post_filter_logic():
    from clip_retrieval_amd.service import KnnHotPath

    hp = KnnHotPath()
    nb = {0: [0, 2], 1: [1], 2: [2, 0, 4], 3: [3], 4: [4, 2], 5: [5, 6], 6: [6, 5]}
    comps = hp.connected_components(nb)
    assert sorted(sorted(c) for c in comps) == [[0, 2, 4], [1], [3], [5, 6]]
    assert [c[0] for c in comps] == [0, 1, 3, 5]  # a component starts at its smallest (= best-ranked) member
    emb = np.eye(4, dtype=np.float32)
    prompts = np.stack([emb[0], emb[1]])  # class 1 ("violent") = direction e1
    assert list(hp.get_violent_items(prompts, emb)) == [1]


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
        nbrs = {i: [int(j) for j in np.flatnonzero(s[i] > thr)] for i in range(emb.shape[0])}
        out = set()
        for g in hp.connected_components(nbrs):
            out |= set(g[1:])
        return out

    R = normalized(x16[np.r_[0:20, 100:110, 499:501, 7]].astype(np.float32))
    assert set(hp.get_non_uniques(R)) == ref_non_uniques(R)
    assert hp.get_non_uniques(R[:1]) == [] and hp.get_non_uniques(np.zeros((0, d), np.float32)) == []

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
