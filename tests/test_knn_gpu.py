"""GPU parity tests of the search half: HIP kernels (through the C ABI / faiss-shaped wrapper) vs the numpy oracle.
Bar (north_star): identical top-k id sets on the same index; scores are fp32 so they are compared to 1e-5 and
ids at the k-th boundary may swap only when their scores differ by < 2e-6 (fp32 summation order)."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NEG = np.float32(-3.4028234663852886e38)


def _data(n, d, seed, normalise=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    if normalise and n:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float16)


def _queries(nq, d, seed, x=None):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    if x is not None and len(x):
        pick = rng.integers(0, len(x), nq)
        q = x[pick].astype(np.float32) + 0.3 * q / np.sqrt(d)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def _check(D, I, Do, Io, ctx, min_exact=0.98):
    from oracle.knn_oracle import topk_sets_equal

    assert D.shape == Do.shape and I.shape == Io.shape and I.dtype == np.int64 and D.dtype == np.float32, ctx
    valid = Io >= 0
    assert np.array_equal(I >= 0, valid), f"{ctx}: -1 padding differs"
    assert (D[~valid] == NEG).all(), f"{ctx}: padding score"
    assert np.allclose(D[valid], Do[valid], rtol=0, atol=1e-5), f"{ctx}: max score err {np.abs(D[valid]-Do[valid]).max()}"
    for i in range(D.shape[0]):
        dv = D[i][I[i] >= 0]
        assert (np.diff(dv) <= 0).all(), f"{ctx}: query {i} not sorted by descending score"
    bad = topk_sets_equal(I, D, Io, Do)
    assert not bad, f"{ctx}: id sets differ beyond near-ties: {bad[:5]}"
    exact = float((I == Io).mean())
    assert exact > min_exact, f"{ctx}: only {exact:.3f} of positions identical"


@pytest.mark.parametrize("n,d", [(1, 256), (31, 768), (33, 768), (1000, 768), (4097, 512), (70001, 768), (5000, 1024)])
def test_flat_topk_parity(n, d):
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    x = _data(n, d, seed=n)
    o = FlatIPOracle(d)
    o.add(x)
    ix = Mi355xIndex(d)
    ix.add(x)
    assert ix.ntotal == n
    for nq, k in [(1, 40), (3, 1), (32, 10), (33, 64), (70, 40)]:
        q = _queries(nq, d, seed=k + nq, x=x)
        D, I = ix.search(q, k)
        Do, Io = o.search(q, k)
        _check(D, I, Do, Io, f"n={n} d={d} nq={nq} k={k}")
    ix.close()


def test_exact_ties_are_ordered_by_id():
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d = 768
    base = _data(300, d, seed=5)
    x = np.concatenate([base, base[:50], base[:50]])  # every one of the first 50 rows appears 3 times
    o, ix = FlatIPOracle(d), Mi355xIndex(d)
    o.add(x)
    ix.add(x)
    q = base[:8].astype(np.float32)
    D, I = ix.search(q, 40)
    Do, Io = o.search(q, 40)
    assert np.array_equal(I, Io), "tied scores must come back in ascending id order"
    assert np.array_equal(D, Do) or np.allclose(D, Do, atol=1e-6)
    for i in range(8):
        assert I[i, :3].tolist() == [i, 300 + i, 350 + i]


def test_wide_scan_64_queries_and_its_fallback():
    """33..64 queries share ONE pass over HBM (fp16-hi scores + exact re-scoring of 64 candidates + proof).  Case 1:
    ordinary data, every proof succeeds.  Case 2: each row exists 100 times, so the 64th approximate score equals the
    k-th exact one, every proof fails and the gated exact scans must deliver the answer.  Both must equal the oracle."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d = 768
    for name, x in (("plain", _data(50000, d, seed=11)), ("dups", np.tile(_data(300, d, seed=12), (100, 1)))):
        o, ix = FlatIPOracle(d), Mi355xIndex(d)
        o.add(x)
        ix.add(x)
        for nq, k in [(64, 40), (40, 10), (33, 48), (64, 1)]:
            q = _queries(nq, d, seed=nq + k, x=x)
            D, I = ix.search(q, k)
            Do, Io = o.search(q, k)
            if name == "dups":
                assert np.array_equal(I, Io), f"{name} nq={nq} k={k}: ties must come back in ascending id order"
                assert np.allclose(D, Do, atol=1e-5)
            else:
                _check(D, I, Do, Io, f"wide {name} nq={nq} k={k}")
        ix.close()


def test_fewer_rows_than_k_and_empty_index():
    from clip_retrieval_amd.knn import Mi355xIndex

    ix = Mi355xIndex(768)
    q = _queries(2, 768, 1)
    D, I = ix.search(q, 5)
    assert (I == -1).all() and (D == NEG).all()
    x = _data(3, 768, 2)
    ix.add(x)
    D, I, R = ix.search_and_reconstruct(q, 8)
    assert (I[:, 3:] == -1).all() and (I[:, :3] >= 0).all() and sorted(I[0, :3].tolist()) == [0, 1, 2]
    assert np.array_equal(R[0, 0], x[I[0, 0]].astype(np.float32))
    assert np.isnan(R[:, 3:]).all()  # faiss memset(-1) pattern


def test_add_in_pieces_f32_and_reconstruct():
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d = 512
    x = _data(5000, d, 9)
    ix, o = Mi355xIndex(d), FlatIPOracle(d)
    ix.add(x[:1234])
    ix.add(x[1234:1300].astype(np.float32))  # f32 rows are rounded to fp16 on the device
    ix.add(x[1300:])
    o.add(x)
    q = _queries(5, d, 3, x)
    D, I, R = ix.search_and_reconstruct(q, 40)
    Do, Io, Ro = o.search_and_reconstruct(q, 40)
    _check(D, I, Do, Io, "pieces")
    assert np.array_equal(R, x[I].astype(np.float32))
    assert np.array_equal(ix.reconstruct(4999), x[4999].astype(np.float32))


def test_dimension_not_multiple_of_256_is_padded():
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d = 640
    x = _data(3000, d, 11)
    ix, o = Mi355xIndex(d), FlatIPOracle(d)
    ix.add(x)
    o.add(x)
    q = _queries(4, d, 5, x)
    D, I, R = ix.search_and_reconstruct(q, 40)
    Do, Io = o.search(q, 40)
    _check(D, I, Do, Io, "d=640")
    assert R.shape == (4, 40, d)


def test_range_search_parity():
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d = 768
    x = _data(20000, d, 21)
    ix, o = Mi355xIndex(d), FlatIPOracle(d)
    ix.add(x)
    o.add(x)
    q = _queries(35, d, 7, x)
    for thr in (0.94, 0.1, 0.05):
        lims, D, I = ix.range_search(q, thr)
        lo, Do, Io = o.range_search(q, thr)
        s = o.scores(q)
        # rows whose score is within fp32 noise of the threshold may fall either side
        for i in range(q.shape[0]):
            a, b = set(I[lims[i]:lims[i + 1]].tolist()), set(Io[lo[i]:lo[i + 1]].tolist())
            for r in a ^ b:
                assert abs(s[i, r] - thr) < 2e-6, (thr, i, r, s[i, r])
            assert (np.diff(I[lims[i]:lims[i + 1]]) > 0).all(), "ids ascending within a query"
        assert lims[0] == 0 and lims[-1] == len(D) == len(I)


def test_range_search_many_queries_on_a_small_index_is_batched():
    """More than 32 queries against a small flat index (the dedup of a request's k result vectors: clip_back.py:290-294): every
    32-query group is launched back to back with its own counters and slice of the hit pool, one synchronisation for all
    (knnx_range_search_once).  Must equal the oracle: sparse results, dense results that overflow the first pool (regrow +
    rescan), hit lists long enough for the radix sort, a ragged last group."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d = 512
    x = _data(3001, d, 33)
    x[1000:1040] = x[5] + 0.02 * _data(40, d, 34)  # a cluster: long lists at a high threshold
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    ix, o = Mi355xIndex(d), FlatIPOracle(d)
    ix.add(x)
    o.add(x)
    xs = x.astype(np.float16).astype(np.float32)  # the stored rows
    for nq, thr in ((3001, 0.94), (333, 0.5), (100, -1.0), (65, 0.02)):
        q = np.ascontiguousarray(xs[:nq])
        lims, D, I = ix.range_search(q, thr)
        lo, Do, Io = o.range_search(q, thr)
        s = o.scores(q)
        assert lims[0] == 0 and lims[-1] == len(D) == len(I)
        for i in range(nq):
            a, b = I[lims[i]:lims[i + 1]], Io[lo[i]:lo[i + 1]]
            if not np.array_equal(a, b):  # rows within fp32 noise of the threshold may fall either side
                for r in set(a.tolist()) ^ set(b.tolist()):
                    assert abs(s[i, r] - thr) < 2e-6, (nq, thr, i, r, s[i, r])
            assert (np.diff(a) > 0).all(), "ids ascending within a query"
            assert np.allclose(D[lims[i]:lims[i + 1]], s[i, a], atol=1e-5)
    lims, D, I = ix.range_search(np.ascontiguousarray(xs[:100]), -1.0)
    assert (np.diff(lims) == 3001).all()  # every row of the index for every query: 300 100 hits through the regrown pool
    ix.close()


@pytest.mark.parametrize("k", [65, 200, 3000])
def test_large_k_parity(k):
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d = 768
    x = _data(20000, d, 31)
    ix, o = Mi355xIndex(d), FlatIPOracle(d)
    ix.add(x)
    o.add(x)
    q = _queries(2, d, 13, x)
    D, I = ix.search(q, k)
    Do, Io = o.search(q, k)
    _check(D, I, Do, Io, f"k={k}")


def test_k_100000_over_a_million_rows():
    """The reference advertises K = 100 000 (README.md:301; clip_back.py:358 special-cases num_result_ids >= 100000); round 3
    refused k > 16 384 (VERDICT r3 missing #2).  Exact id lists against the numpy oracle at k = 100 000 over 1 M rows (the
    density-extrapolated descent: N < 128 k), at the new limit 131 072 with the stored rows reconstructed, and k beyond it refused."""
    from clip_retrieval_amd import HipLibraryError
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d, n = 256, 1_000_000
    x = _data(n, d, 77)
    ix, o = Mi355xIndex(d), FlatIPOracle(d)
    for lo in range(0, n, 250_000):
        ix.add(x[lo:lo + 250_000])
    o.add(x)
    q = _queries(2, d, 5, x)
    for k in (100_000, 131_072):
        D, I = ix.search(q, k)
        Do, Io = o.search(q, k)
        # 100 k of 1 M rows: neighbouring scores are ~1e-6 apart, so the numpy oracle's f32 summation order swaps a few per cent of
        # adjacent POSITIONS; the id sets (beyond 2e-6 ties), the scores (1e-5) and the order of ours are held exactly
        _check(D, I, Do, Io, f"k={k} over 1 M rows", min_exact=0.9)
        assert (np.diff(D, axis=1) <= 0).all()
        moved = I != Io
        assert np.abs(D[moved] - Do[moved]).max() < 2e-6  # every position that differs is such a near-tie
    D, I, R = ix.search_and_reconstruct(q[:1], 100_000)
    assert R.shape == (1, 100_000, d) and np.array_equal(R[0, ::997], x[I[0, ::997]].astype(np.float32))
    with pytest.raises(HipLibraryError):
        ix.search(q, 131_073)
    ix.close()


def test_k_100000_takes_the_sampled_threshold_on_a_large_index():
    """N >= 128 k: ONE range scan above the j-th best score of a strided sample (every 3 126-th tile at k = 100 000).  16 M x 256
    synthetic rows, ids checked against chunked torch matmul + topk on the same bytes (independent arithmetic)."""
    import torch

    from clip_retrieval_amd.knn import Mi355xIndex

    d, n, k = 256, 16_000_000, 100_000
    X = torch.empty((n, d), dtype=torch.float16, device="cuda")  # the arena is a torch tensor the index borrows (as bench.py does)
    ix = Mi355xIndex(d)
    ix.attach_device_rows(X.data_ptr(), n)
    ix.synth_fill(n, 21)
    q = _queries(1, d, 8)
    q = ix.reconstruct_batch(np.asarray([12345], dtype=np.int64)).astype(np.float32) * 0.7 + 0.3 * q
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    D, I = ix.search(q, k)
    assert (I >= 0).all() and len(set(I[0].tolist())) == k and (np.diff(D[0]) <= 0).all() and I[0, 0] == 12345
    qt = torch.from_numpy(q).cuda()
    best_s, best_i = None, None
    for lo in range(0, n, 2_000_000):
        ts, ti = torch.topk((qt @ X[lo:lo + 2_000_000].float().T)[0], k)
        ti = ti + lo
        best_s, best_i = (ts, ti) if best_s is None else (torch.cat([best_s, ts]), torch.cat([best_i, ti]))
        best_s, sel = torch.topk(best_s, k)
        best_i = best_i[sel]
    want_s, want_i = best_s.cpu().numpy(), best_i.cpu().numpy()
    assert np.allclose(D[0], want_s, atol=2e-6)
    assert len(set(I[0].tolist()) ^ set(want_i.tolist())) <= 4  # only exact-tie / 1e-7 swaps at the k-th boundary
    ix.close()


def test_synthetic_corpus_is_bit_identical_to_the_cpu_derivation():
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import synth_rows

    d, n, seed = 768, 200_000, 3
    ix = Mi355xIndex(d)
    ix.synth_fill(n, seed)
    rows = np.array([0, 1, 63, 64, 4097, 199_999], dtype=np.int64)
    got = ix.reconstruct_batch(rows)
    want = synth_rows(rows, d, seed).astype(np.float32)
    assert np.array_equal(got, want)


def test_full_scale_properties_planted_neighbours():
    """Size-independent checks on a corpus the oracle cannot scan in seconds (4 M x 768 = 6 GB):
    every planted neighbour is the top hit, scores agree with an fp32 recomputation of the returned rows,
    and results are independent of how queries are batched."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import planted_queries, synth_rows

    d, n, seed = 768, 4_000_000, 3
    ix = Mi355xIndex(d)
    ix.synth_fill(n, seed)
    rng = np.random.default_rng(0)
    planted = np.sort(rng.choice(n, 48, replace=False))
    planted[0], planted[-1] = 0, n - 1
    q = planted_queries(planted, d, seed)
    D, I = ix.search(q, 40)
    assert np.array_equal(I[:, 0], planted)
    assert (np.diff(D, axis=1) <= 0).all() and (I >= 0).all() and (I < n).all()
    for i in (0, 17, 47):  # recompute the scores of the returned rows from the CPU derivation of the corpus
        rows = synth_rows(I[i], d, seed).astype(np.float32)
        assert np.allclose(rows @ q[i], D[i], atol=1e-5)
        assert len(set(I[i].tolist())) == 40
    D1, I1 = ix.search(q[:1], 40)
    assert np.array_equal(I1[0], I[0]) and np.allclose(D1[0], D[0], atol=1e-6)


def test_concurrent_single_query_callers_are_coalesced_correctly():
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d = 768
    x = _data(30000, d, 41)
    ix, o = Mi355xIndex(d), FlatIPOracle(d)
    ix.add(x)
    o.add(x)
    q = _queries(24, d, 17, x)
    Do, Io = o.search(q, 40)
    out, errs = [None] * 24, []

    def call(i):
        try:
            out[i] = ix.search_and_reconstruct(q[i:i + 1], 40)
        except Exception as e:  # pylint: disable=broad-except
            errs.append(e)

    ts = [threading.Thread(target=call, args=(i,)) for i in range(24)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for i in range(24):
        D, I, R = out[i]
        _check(D, I, Do[i:i + 1], Io[i:i + 1], f"thread {i}")
        assert np.array_equal(R[0], x[I[0]].astype(np.float32))


def test_native_coalescer_serves_many_threads_from_few_scans():
    """Round 4 (VERDICT r3 missing #3; SURVEY 8b "knnx_search is re-entrant; internally a batching queue"): 96 threads x 4 calls
    of n = 1 with mixed k and with / without reconstruction go through the library's own queue.  Every answer equals the
    uncoalesced one (same ids, scores to f32 summation order), errors reach the thread that made the bad call (thread-local message), the queue served the
    384 + calls in far fewer scans, and an index with coalescing off answers the same."""
    from clip_retrieval_amd import HipLibraryError
    from clip_retrieval_amd.knn import Mi355xIndex

    d = 512
    x = _data(60000, d, 43)
    ix, plain = Mi355xIndex(d), Mi355xIndex(d, coalesce=False)
    ix.add(x)
    plain.add(x)
    q = _queries(96, d, 19, x)
    want = {k: plain.search_and_reconstruct(q, k) for k in (40, 7)}  # one batched call each: rows do not depend on their batch
    errs, bad = [], []
    b0 = ix.coalesce_stats()

    def call(i):
        try:
            for rep in range(4):
                k = 40 if (i + rep) % 3 else 7
                if rep % 2:
                    D, I, R = ix.search_and_reconstruct(q[i:i + 1], k)
                    assert np.array_equal(R, want[k][2][i:i + 1])
                else:
                    D, I = ix.search(q[i:i + 1], k)
                # same ids; the scores agree to f32 summation order (which scan kernel serves a query depends on how many
                # queries share its pass -- 32-query exact, 64-query wide + re-score -- and they sum in different orders)
                assert np.array_equal(I, want[k][1][i:i + 1]) and np.allclose(D, want[k][0][i:i + 1], rtol=0, atol=1e-6)
            if i % 16 == 0:
                try:
                    ix.search(q[i:i + 1], 200_000)
                except HipLibraryError as e:
                    bad.append(str(e))
        except Exception as e:  # pylint: disable=broad-except
            errs.append(repr(e))

    ts = [threading.Thread(target=call, args=(i,)) for i in range(96)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:3]
    assert len(bad) == 6 and all("131072" in m for m in bad)
    batches, queries, largest = (a - b for a, b in zip(ix.coalesce_stats(), b0))
    # (how many callers share a scan depends on how fast Python starts threads against a 50 us scan: "some" is what is asserted)
    assert queries == 96 * 4 and batches < queries and ix.coalesce_stats()[2] >= 2, (batches, queries, ix.coalesce_stats())
    assert plain.coalesce_stats() == (0, 0, 0)
    ix.close()
    plain.close()


@pytest.mark.parametrize("ivf", [False, True])
def test_search_dedup_links_are_the_reference_links(ivf):
    """knnx_search_dedup: the top-k of one query plus the links of clip_back.py:290-309 (`IndexFlatIP(R).range_search(R, 0.94)` on
    the normalised result vectors) computed on the device for the whole coalesced batch.  Index with planted near-duplicate
    groups; links compared with the numpy statement of the reference on the reconstructed rows, from 48 concurrent threads;
    flat and IVF (inverse id map); a short answer (-1 padding) and want_r."""
    from clip_retrieval_amd.knn import Mi355xIndex, build_ivf_index
    from clip_retrieval_amd.service import KnnHotPath, normalized

    rng = np.random.default_rng(5)
    d, n = 768, 20000
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    for g in range(40):  # groups of 2 .. 5 near-copies, scaled differently (the reference normalises the results first)
        src = 100 * g
        for c in range(1, int(rng.integers(2, 6))):
            x[src + c] = (x[src] + rng.uniform(0.002, 0.02) * rng.standard_normal(d).astype(np.float32)) * rng.uniform(0.5, 2.0)
    x16 = x.astype(np.float16)
    if ivf:
        ix = build_ivf_index(x16, 32, nprobe=32, niter=2)
    else:
        ix = Mi355xIndex(d)
        ix.add(x16)
    qs = np.stack([x[100 * g] / np.linalg.norm(x[100 * g]) for g in range(40)] + [x[7 + i] for i in range(8)]).astype(np.float32)

    def ref_links(I):
        R = normalized(x16[I].astype(np.float32))
        s = R @ R.T
        return [(i, j) for i in range(len(I)) for j in range(i + 1, len(I)) if s[i, j] > 0.94], s

    out, errs = {}, []

    def call(t):
        try:
            out[t] = ix.search_dedup(qs[t:t + 1], 40, 0.94, want_r=(t % 2 == 0))
        except Exception as e:  # pylint: disable=broad-except
            errs.append(repr(e))

    ts = [threading.Thread(target=call, args=(t,)) for t in range(len(qs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:3]
    D0, I0 = ix.search(qs, 40)
    n_links = 0
    for t in range(len(qs)):
        D, I, R, links = out[t]
        assert np.array_equal(I, I0[t:t + 1]) and np.allclose(D, D0[t:t + 1], rtol=0, atol=1e-6)  # (scores: f32 summation order of the scan kernel)
        assert (R is None) == (t % 2 == 1)
        if R is not None:
            assert np.array_equal(R[0], x16[I[0]].astype(np.float32))
        want, s = ref_links(I[0])
        got = [tuple(int(v) for v in p) for p in links]
        near = {(i, j) for i in range(40) for j in range(i + 1, 40) if abs(s[i, j] - 0.94) < 1e-5}  # f32 summation order
        assert set(got) - near == set(want) - near and got == sorted(got), (t, got, want)
        if not near:  # the groups that follow from the links are the reference's (its get_non_uniques keeps the smallest of a group)
            assert KnnHotPath.non_uniques_from_pairs(links, 40) == KnnHotPath.non_uniques_from_pairs(np.asarray(want, dtype=np.int32).reshape(-1, 2), 40)
        n_links += len(got)
    assert n_links > 60  # the planted groups were found
    # a short answer: k > rows reachable -> -1 padding takes no part in the links
    small = Mi355xIndex(d)
    small.add(x16[100:104])
    D, I, R, links = small.search_dedup(qs[1:2], 10, 0.94, want_r=True)
    assert (I[0, 4:] == -1).all() and all(max(p) < 4 for p in links)
    assert {tuple(int(v) for v in p) for p in links} == set(ref_links(I[0, :4])[0])
    small.close()
    ix.close()


def test_bad_arguments_raise_like_faiss():
    from clip_retrieval_amd import HipLibraryError
    from clip_retrieval_amd.knn import Mi355xIndex

    ix = Mi355xIndex(768)
    with pytest.raises(TypeError):
        ix.search(np.zeros((1, 768), np.float64), 4)
    with pytest.raises(AssertionError):
        ix.search(np.zeros((1, 512), np.float32), 4)
    with pytest.raises(HipLibraryError):
        ix.search(np.zeros((1, 768), np.float32), 200_000)  # beyond KNNX_MAX_K = 131 072
    D, I = ix.search(np.zeros((1, 768), np.float32), 20000)  # (an empty index answers -1 / -FLT_MAX at any allowed k)
    assert (I == -1).all() and (D == NEG).all()


# ------------------------------------------------------------------------------------------ IVF-Flat (BASELINE config 5)
@pytest.mark.parametrize("n,d,nlist,nprobe", [(20000, 768, 64, 8), (5000, 1024, 16, 16), (3000, 512, 100, 1), (40, 768, 8, 3)])
def test_ivf_flat_parity(n, d, nlist, nprobe):
    """IVF search = exactly the top-k of the rows in the nprobe best lists (faiss IndexIVFFlat semantics), compared with
    the numpy oracle on the SAME centroids and assignment; nprobe = nlist must reproduce the flat result."""
    from clip_retrieval_amd.knn import Mi355xIndex, build_ivf_index
    from oracle.knn_oracle import FlatIPOracle, IVFFlatOracle

    x = _data(n, d, seed=n + nlist)
    rng = np.random.default_rng(nlist)
    cent = x[rng.choice(n, nlist, replace=False)]  # fixed centroids: the build's k-means is not what is under test here
    ix = build_ivf_index(x, nlist, nprobe=nprobe, centroids=cent)
    assert ix.ntotal == n and ix.nlist == nlist
    o = IVFFlatOracle(d, cent, ix.ivf_lists, x)
    # more than 32 queries = ONE multi-block pass of up to 256 (round 6: knnx_api.hip scan_topk_ivf_multi); 300 = a pass of 256 + one of 44
    for nq, k in [(1, 40), (7, 10), (32, 64), (45, 40), (64, 40), (200, 20), (256, 40), (300, 64)]:
        q = _queries(nq, d, seed=k + nq, x=x)
        D, I = ix.search(q, k)
        Do, Io = o.search(q, k, nprobe)
        _check(D, I, Do, Io, f"ivf n={n} d={d} nlist={nlist} nprobe={nprobe} nq={nq} k={k}")
    # reconstruct goes through the inverse id map
    q = _queries(3, d, seed=3, x=x)
    D, I, R = ix.search_and_reconstruct(q, 5)
    ok = I >= 0
    assert np.array_equal(R[ok], x[I[ok]].astype(np.float32))
    if nlist <= 64:
        ix.nprobe = nlist
        f = FlatIPOracle(d)
        f.add(x)
        D, I = ix.search(q, 40)
        Do, Io = f.search(q, 40)
        _check(D, I, Do, Io, "ivf with nprobe = nlist equals flat")
    ix.close()


@pytest.mark.parametrize("nprobe", [3, 80])
def test_ivf_range_search_and_large_k(nprobe):
    """IVF indexes also serve `range_search` (clip_filter.py:52, the dedup of clip_back.py:290-309) and k > 64 (the
    front end asks for 3000 results, clip_back.py:358): both over exactly the rows of the nprobe best lists.  A k larger than
    what the probed lists hold is padded with -1 / -FLT_MAX like faiss does."""
    from clip_retrieval_amd.knn import build_ivf_index
    from oracle.knn_oracle import IVFFlatOracle

    n, d, nlist = 30_011, 512, 96
    x = _data(n, d, seed=77)
    rng = np.random.default_rng(5)
    cent = x[rng.choice(n, nlist, replace=False)]
    ix = build_ivf_index(x, nlist, nprobe=nprobe, centroids=cent)
    o = IVFFlatOracle(d, cent, ix.ivf_lists, x)
    q = _queries(5, d, seed=9, x=x)
    for thr in (0.3, 0.08, -2.0):  # -2: every row of the probed lists
        lims, D, I = ix.range_search(q, thr)
        lo, Do, Io = o.range_search(q, thr, nprobe)
        assert np.array_equal(lims, lo), f"thr={thr}: {lims} vs {lo}"
        for i in range(5):
            a, b = slice(lims[i], lims[i + 1]), slice(lo[i], lo[i + 1])
            oa, ob = np.argsort(I[a]), np.argsort(Io[b])
            assert np.array_equal(I[a][oa], Io[b][ob]), f"thr={thr} query {i}: id sets differ"
            assert np.abs(D[a][oa] - Do[b][ob]).max(initial=0.0) < 1e-5
    for k in (200, 1000):
        D, I = ix.search(q[:3], k)
        Do, Io = o.search(q[:3], k, nprobe)
        _check(D, I, Do, Io, f"ivf nprobe={nprobe} k={k}")
    if nprobe == 3:  # three lists of ~300 rows hold fewer than 3000 rows: the tail is padding
        D, I = ix.search(q[:2], 3000)
        Do, Io = o.search(q[:2], 3000, nprobe)
        assert (Io[:, -1] == -1).all()
        _check(D, I, Do, Io, "ivf k = 3000 beyond the probed lists")
    ix.close()


def test_ivf_trained_recall():
    """k-means centroids from the library's own assignment scan: recall@10 vs exact flat on clustered data."""
    from clip_retrieval_amd.knn import build_ivf_index
    from oracle.knn_oracle import FlatIPOracle

    rng = np.random.default_rng(0)
    d, nclu = 256, 32
    centers = rng.standard_normal((nclu, d)).astype(np.float32)
    x = centers[rng.integers(0, nclu, 30000)] + 0.35 * rng.standard_normal((30000, d)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
    ix = build_ivf_index(x, nlist=32, nprobe=4, niter=6)
    f = FlatIPOracle(d)
    f.add(x)
    q = _queries(64, d, seed=1, x=x)
    _, I = ix.search(q, 10)
    _, Io = f.search(q, 10)
    recall = np.mean([len(set(I[i]) & set(Io[i])) / 10 for i in range(64)])
    assert recall > 0.9, recall
    ix.close()


# ---------------------------------------------------------------------------------------------------------------
# RQ scan: up to 256 queries per pass (csrc/knn_rq_kernels.hip).  KNNX_RQ_MIN_ROWS=0 makes small indexes take the path.
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(params=["fp16", "int8"])
def rq_on_small_indexes(monkeypatch, request):
    """Both forms of the register-stationary pass: the fp16 scan (KNNX_I8=0) and the int8 first stage in front of it (default)."""
    monkeypatch.setenv("KNNX_RQ_MIN_ROWS", "0")  # read by knnx_create
    monkeypatch.setenv("KNNX_I8", "0" if request.param == "fp16" else "1")
    return request.param


@pytest.mark.parametrize("d", [768, 512, 1024])
def test_rq_scan_256_queries_parity(rq_on_small_indexes, d):
    """65..256 queries share ONE pass over HBM: queries stationary in registers, thresholds from a strided sample,
    hits re-scored exactly, exactness proven per query.  Must equal the oracle like every other path; the counters
    show that the proof-based path (not the 32-query scan) served them."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    n = 60_000 + 13  # ragged last tile
    x = _data(n, d, seed=21)
    o, ix = FlatIPOracle(d), Mi355xIndex(d)
    o.add(x)
    ix.add(x)
    served0 = ix.stats()[0]
    for nq, k in [(256, 40), (200, 10), (65, 48), (129, 1), (300, 40)]:
        q = _queries(nq, d, seed=nq + k, x=x)
        D, I = ix.search(q, k)
        Do, Io = o.search(q, k)
        _check(D, I, Do, Io, f"rq d={d} nq={nq} k={k}")
    served, failed = ix.stats()
    assert served - served0 >= 900, "the proof-based scans did not serve these batches"
    assert failed <= 8, f"{failed} proofs failed on ordinary data"  # small index: the sample is the whole index, hits = k + 8
    assert (ix.i8_served() >= 900) == (rq_on_small_indexes == "int8"), "the wrong form of the pass served these batches"
    ix.close()


def test_rq_scan_ties_floods_and_fallback(rq_on_small_indexes):
    """Adversarial data for the RQ path.  (1) every row exists 100 times: the k-th exact score ties with rows at the
    threshold, proofs fail, the gated exact scans must answer (ids in ascending order among ties).  (2) 40 000 copies of
    ONE row: every copy reaches the threshold, the per-query hit list (16 384) overflows -> fallback.  Both must equal
    the oracle exactly."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d = 768
    base = _data(300, d, seed=12)
    flood = np.concatenate([np.tile(base[:1], (40_000, 1)), _data(5_000, d, seed=13)])
    for name, x in (("dups", np.tile(base, (100, 1))), ("flood", flood)):
        o, ix = FlatIPOracle(d), Mi355xIndex(d)
        o.add(x)
        ix.add(x)
        for nq, k in [(256, 40), (100, 5)]:
            q = _queries(nq, d, seed=nq + k, x=x)
            if name == "flood":
                q[: nq // 2] = base[:1].astype(np.float32)  # half of the queries ARE the flooding row
            D, I = ix.search(q, k)
            Do, Io = o.search(q, k)
            assert np.array_equal(I, Io), f"{name} nq={nq} k={k}: ids (ties in ascending id order)"
            assert np.allclose(D, Do, atol=1e-5)
        if rq_on_small_indexes == "fp16":
            assert ix.stats()[1] > 0, f"{name}: expected failed proofs (fallback path untested otherwise)"
        # (int8: its proof is about completeness only -- exact ties at the threshold are ordinary hits, so the duplicated corpus needs
        # no fallback there; the flood overflows its 32 768-entry list as well; test_i8_first_stage_overflow_falls_back asserts that path)
        ix.close()


def test_rq_full_scale_properties_256_planted_queries():
    """8 M x 768 (12 GB; the RQ path's natural size class): 256 planted queries in one call -- every planted neighbour is
    the top hit, scores agree with an fp32 recomputation from the CPU derivation of the corpus, and the answers are the
    same as the 32-query exact scan's (batching independence across scan kernels)."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import planted_queries, synth_rows

    d, n, seed = 768, 8_000_000, 3
    ix = Mi355xIndex(d)
    ix.synth_fill(n, seed)
    rng = np.random.default_rng(1)
    planted = np.sort(rng.choice(n, 256, replace=False))
    planted[0], planted[-1] = 0, n - 1
    q = planted_queries(planted, d, seed)
    s0 = ix.stats()
    D, I = ix.search(q, 40)
    s1 = ix.stats()
    assert s1[0] - s0[0] == 256, "256 queries must have gone through one RQ pass"
    assert s1[1] - s0[1] == 0, "no proof may fail on this corpus"
    assert np.array_equal(I[:, 0], planted)
    assert (np.diff(D, axis=1) <= 0).all() and (I >= 0).all() and (I < n).all()
    for i in (0, 100, 255):
        rows = synth_rows(I[i], d, seed).astype(np.float32)
        assert np.allclose(rows @ q[i], D[i], atol=1e-5)
        assert len(set(I[i].tolist())) == 40
    for lo in (0, 224):  # the same queries through the exact 32-query scan
        D32, I32 = ix.search(q[lo:lo + 32], 40)
        assert np.array_equal(I32, I[lo:lo + 32]) and np.allclose(D32, D[lo:lo + 32], atol=2e-6)
    ix.close()


def test_i8_first_stage_equals_exact_scans(monkeypatch):
    """The int8 first stage (default on flat indexes >= 2^21 rows) against the same index with it turned off: 3 M x 768 synthetic
    rows, batches of 1, 7, 32, 64, 200 and 300 planted + random queries -- ids identical, scores identical (both re-score the same
    fp16 rows with the same arithmetic), no fallback on this corpus, and the counters say which path served."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import planted_queries

    d, n, seed = 768, 3_000_000, 5
    rng = np.random.default_rng(2)
    planted = np.sort(rng.choice(n, 300, replace=False))
    q = planted_queries(planted, d, seed)
    q[1::3] = rng.standard_normal((len(q[1::3]), d)).astype(np.float32)  # a third of the queries are not near any row
    q[2::7] *= 3.7  # ... and some are not unit vectors
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("KNNX_I8", mode)
        ix = Mi355xIndex(d)
        ix.synth_fill(n, seed)
        out = []
        for nq in (1, 7, 32, 64, 200, 300):
            out.append(ix.search(q[:nq], 40))
        served, failed = ix.stats()
        res[mode] = (out, ix.i8_served(), failed)
        ix.close()
    assert res["0"][1] == 0 and res["1"][1] == 1 + 7 + 32 + 64 + 200 + 300, (res["0"][1], res["1"][1])
    assert res["1"][2] == 0, f"{res['1'][2]} int8 hit lists overflowed on an ordinary corpus"
    for (D0, I0), (D1, I1) in zip(res["0"][0], res["1"][0]):
        _check(D1, I1, D0, I0, f"int8 first stage vs exact scans, {len(D0)} queries", min_exact=0.999)
        assert (I0 == I1).mean() > 0.999, "ids differ between the int8 first stage and the exact scans beyond a near-tie or two"
    assert np.array_equal(res["1"][0][-1][1][0::3, 0], planted[0::3])  # planted neighbours (the rows of the unperturbed queries)


def test_i8_first_stage_after_the_rows_change(monkeypatch):
    """The int8 copy follows the rows: rows appended while the index has less than doubled are quantised with the existing column
    scales (second add below), beyond that everything is redone (third add).  Results must equal the oracle at every stage -- the
    new rows here are 3 x larger than the ones the scales were taken over, so most of their components clamp and the bound has to
    carry that."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    monkeypatch.setenv("KNNX_RQ_MIN_ROWS", "0")
    d = 512
    x = _data(55_000, d, seed=31)
    x[20_000:30_000] *= 3  # (not unit vectors: the appended rows exceed the existing column scales)
    o, ix = FlatIPOracle(d), Mi355xIndex(d)
    for part in (x[:20_000], x[20_000:30_000], x[30_000:]):
        o.add(part)
        ix.add(part)
        q = _queries(40, d, seed=len(part), x=x[: o.ntotal])
        D, I = ix.search(q, 10)
        Do, Io = o.search(q, 10)
        _check(D, I, Do, Io, f"int8 after add ntotal={o.ntotal}")
    assert ix.i8_served() == 120
    ix.close()


def test_i8_first_stage_forms_for_dominant_columns(monkeypatch):
    """Embeddings with a few dominant dimensions (two columns 6 x the rest plus a common offset, as CLIP embeddings have): the column
    scales differ widely.  Round 5: the library keeps ONE int8 plane, moves the dominant columns to the front of its private copy and
    gives the queries 14-bit digits there (knnx_i8_dominant) -- results equal the oracle without a single fallback, every batch size
    on the int8 path.  With that form off (KNNX_I8_DOM=0) it picks TWO planes (round 4), again without fallbacks; with one plain plane
    forced the same data is still answered exactly -- through wider hit lists.  An isotropic index: one plane, no dominant column."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    monkeypatch.setenv("KNNX_RQ_MIN_ROWS", "0")
    d, n = 768, 150_000
    rng = np.random.default_rng(51)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[:, :2] = 6.0 * x[:, :2] + 3.0
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x = x.astype(np.float16)
    o = FlatIPOracle(d)
    o.add(x)
    q = _queries(200, d, seed=52, x=x)
    Do, Io = o.search(q, 40)
    for env, planes, dom in (({}, 1, [0, 1]), ({"KNNX_I8_DOM": "0"}, 2, []), ({"KNNX_I8_PLANES": "1"}, 1, [])):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        ix = Mi355xIndex(d)
        ix.add(x)
        for lo, hi in ((0, 1), (1, 41), (41, 200)):
            D, I = ix.search(q[lo:hi], 40)
            _check(D, I, Do[lo:hi], Io[lo:hi], f"dominant columns, {env}, queries {lo}:{hi}")
        # (the 159-query batch: with two planes more than 128 queries go to the fp16 register-stationary pass)
        assert ix.i8_planes() == planes and ix.i8_dominant() == dom
        assert ix.i8_served() == (41 if planes == 2 else 200)
        if planes == 2 or dom:
            assert ix.stats()[1] == 0, f"{env}: must not need the fallback on this corpus"
        ix.close()
        for k_ in env:
            monkeypatch.delenv(k_)
    iso = Mi355xIndex(d)
    iso.add(_data(60_000, d, seed=53))
    iso.search(_queries(3, d, seed=54), 5)
    assert iso.i8_planes() == 1 and iso.i8_dominant() == []
    iso.close()


@pytest.mark.parametrize("d,n,cols", [(512, 70_003, [5]), (768, 90_017, [700, 2, 63]), (1024, 64_000, [1023, 0, 511, 512]),
                                      (768, 50_000, [1, 2, 3, 64, 65])])
def test_i8_dominant_columns_anywhere_in_the_row(monkeypatch, d, n, cols):
    """The dominant-column form of the int8 first stage (include/knnx.h, knnx_i8_dominant): the columns may sit anywhere -- the copy
    permutes them to its first bytes (a dominant column that already sits among the first four, and one that has to swap with another
    dominant column's target, included) --, rows may be appended afterwards (quantised with the same permutation), the copy may be
    partial, the last tile ragged; five dominant columns are one too many for the form and get two planes.  Every batch size class
    equals the oracle, served by the int8 path without fallbacks."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle, Int8FirstStage

    monkeypatch.setenv("KNNX_RQ_MIN_ROWS", "0")
    if d == 768 and len(cols) == 3:
        monkeypatch.setenv("KNNX_I8_MAX_BYTES", str(60_000 * d))  # a partial copy: rows behind it go through the fp16 pass
    rng = np.random.default_rng(d + len(cols))
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[:, cols] = 6.0 * x[:, cols] + 3.0 * rng.choice([-1.0, 1.0], size=len(cols)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x = x.astype(np.float16)
    n0 = n - 4_001  # the rest is appended after the first search
    o = FlatIPOracle(d)
    o.add(x[:n0])
    ix = Mi355xIndex(d)
    ix.add(x[:n0])
    want_planes, want_dom = Int8FirstStage(x[:n0]).form()
    assert (want_planes, want_dom) == ((1, sorted(cols)) if len(cols) <= 4 else (2, []))
    served = 0
    for step in range(2):
        nq_all = 300 if want_planes == 1 else 100
        q = _queries(nq_all, d, seed=60 + step, x=x[: o.ntotal])
        # neighbours planted in the last rows (ragged tile; after the append: appended rows)
        q[:4] = x[o.ntotal - 4: o.ntotal].astype(np.float32) + 0.01 * rng.standard_normal((4, d)).astype(np.float32)
        Do, Io = o.search(q, 10)
        for lo, hi in ((0, 1), (1, 34), (34, 100), (100, nq_all)):
            if hi <= lo:
                continue
            D, I = ix.search(q[lo:hi], 10)
            _check(D, I, Do[lo:hi], Io[lo:hi], f"dominant {cols} d={d} step {step} queries {lo}:{hi}")
        served += nq_all
        assert set(Io[:4, 0]) == set(range(o.ntotal - 4, o.ntotal))
        assert ix.i8_planes() == want_planes and sorted(ix.i8_dominant()) == want_dom
        assert ix.i8_served() == served and ix.stats()[1] == 0
        if step == 0:
            o.add(x[n0:])
            ix.add(x[n0:])
    if d == 768 and len(cols) == 3:
        assert ix.i8_rows() == 60_000 - 60_000 % 32
    ix.close()


def test_i8_dominant_digits_clamp_and_dominant_only_queries_stay_exact(monkeypatch):
    """Edges of the dominant-column form (tests/test_oracle.py::test_int8_dominant_digits_edge_cases_keep_the_bound has the arithmetic):
    one column ~200 x the rest, so that the queries' components there exceed 14 bits and clamp, and queries that live only in that column
    (no scale from the others: since round 6 they take it from the dominant components' 14-bit range -- before, every digit rounded to
    zero, every row was admitted and the gated exact scan answered).  Results equal the oracle either way; the bound may cost
    fallbacks here, never a wrong id."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    monkeypatch.setenv("KNNX_RQ_MIN_ROWS", "0")
    d, n = 768, 80_000
    rng = np.random.default_rng(91)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[:, 7] = 50.0 * x[:, 7] + 30.0
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x = x.astype(np.float16)
    o = FlatIPOracle(d)
    o.add(x)
    ix = Mi355xIndex(d)
    ix.add(x)
    q = _queries(70, d, seed=92, x=x)
    q[:60, 7] *= 4.0  # (queries are not normalised by the index: four times the corpus' own component takes the digits past 14 bits)
    q[60:] = 0.0
    q[60:, 7] = np.linspace(-1.0, 1.0, 10, dtype=np.float32)
    q[64, 7] = 0.0  # ... and the zero vector
    Do, Io = o.search(q, 10)
    for lo, hi in ((0, 1), (1, 60), (60, 70), (0, 70)):
        D, I = ix.search(q[lo:hi], 10)
        m = min(hi, 60)  # queries [lo, m) have all their components, [max(lo, 60), hi) only the dominant one
        if lo < m:
            _check(D[: m - lo], I[: m - lo], Do[lo:m], Io[lo:m], f"clamped digits, queries {lo}:{m}")
        # the dominant-only queries score every row by ONE column: thousands of ties within rounding -- compare the scores
        if hi > 60:
            np.testing.assert_allclose(D[max(60, lo) - lo:], Do[max(lo, 60):hi], rtol=0, atol=2e-6)
    assert ix.i8_dominant() == [7] and ix.i8_served() == 1 + 59 + 10 + 70
    ix.close()


@pytest.mark.parametrize("d,n,budget_rows", [(768, 90_001, 40_000), (1024, 70_013, 30_000), (512, 60_000, 59_999)])
def test_i8_partial_copy_int8_part_plus_fp16_rest_equals_the_oracle(monkeypatch, d, n, budget_rows):
    """BASELINE's headline shard (125 M x 768 fp16 = 192 GB per GPU) leaves no room for a whole int8 copy, so the copy may be PARTIAL
    (include/knnx.h, knnx_i8_rows): the leading rows get the int8 first stage, the rows behind them the fp16 register-stationary pass,
    both into one hit list with one proof.  Here the budget is made artificially small (KNNX_I8_MAX_BYTES); neighbours are planted on
    both sides of the boundary and in the ragged last tile; every batch size class (1, <= 64, <= 128, 256, > 256) must equal the oracle
    and be served by the int8 path."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    monkeypatch.setenv("KNNX_RQ_MIN_ROWS", "0")
    monkeypatch.setenv("KNNX_I8_MAX_BYTES", str(budget_rows * d))
    x = _data(n, d, seed=61)
    o, ix = FlatIPOracle(d), Mi355xIndex(d)
    o.add(x)
    ix.add(x)
    n8 = (budget_rows // 32) * 32
    rows = np.minimum(np.r_[0, 17, n8 - 1, n8, n8 + 1, n8 + 33, n - 1, n - 2, np.random.default_rng(62).integers(0, n, 292)], n - 1)
    q = x[rows].astype(np.float32) + 0.02 * np.random.default_rng(63).standard_normal((len(rows), d)).astype(np.float32)
    Do, Io = o.search(q, 40)
    served = 0
    for lo, hi in ((0, 1), (1, 8), (8, 70), (70, 190), (0, 256), (0, 300)):
        D, I = ix.search(q[lo:hi], 40)
        _check(D, I, Do[lo:hi], Io[lo:hi], f"partial int8 copy d={d} queries {lo}:{hi}")
        served += hi - lo
    assert ix.i8_rows() == n8, (ix.i8_rows(), n8)
    assert ix.i8_served() == served and ix.stats()[1] == 0, (ix.i8_served(), served, ix.stats())
    assert np.array_equal(Io[:6, 0], rows[:6])  # the planted rows around the boundary are the top hits
    # the budget grows: the next rebuild takes the whole index, nothing changes in the results
    ix.close()


def test_i8_tile_ordered_copy_ragged_sizes_and_appends(monkeypatch):
    """The int8 copy is stored tile-ordered (the LDS image of the scan, knn_i8_quant_kernel): row counts that end inside a 16-row half
    tile, inside the second half, and exactly on a tile; rows appended so that the new rows share a half tile with old ones."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    monkeypatch.setenv("KNNX_RQ_MIN_ROWS", "0")
    d = 768
    x = _data(40_000, d, seed=71)
    for sizes in ((33_001,), (32_016 + 7, 5), (32_000, 1, 14, 17), (20_005, 9_990, 3)):
        o, ix = FlatIPOracle(d), Mi355xIndex(d)
        at = 0
        for sz in sizes:
            o.add(x[at:at + sz])
            ix.add(x[at:at + sz])
            at += sz
            rows = np.r_[at - 1, max(0, at - sz), max(0, at - sz - 1), np.random.default_rng(at).integers(0, at, 29)]
            q = x[rows].astype(np.float32) + 0.02 * np.random.default_rng(at + 1).standard_normal((32, d)).astype(np.float32)
            D, I = ix.search(q, 10)
            Do, Io = o.search(q, 10)
            _check(D, I, Do, Io, f"tile-ordered int8 copy, sizes {sizes} at {at}")
            assert ix.i8_rows() == at
        assert ix.stats()[1] == 0
        ix.close()


def test_i8_first_stage_overflow_falls_back(monkeypatch):
    """200 000 copies of one row and queries equal to it: every copy passes the int8 admission, the 32 768-entry hit list
    overflows, the query is answered by the gated exact scan -- ids in ascending order among the ties."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    monkeypatch.setenv("KNNX_RQ_MIN_ROWS", "0")
    d = 768
    base = _data(1, d, seed=41)
    x = np.concatenate([_data(3_000, d, seed=42), np.tile(base, (200_000, 1))])
    o, ix = FlatIPOracle(d), Mi355xIndex(d)
    o.add(x)
    ix.add(x)
    q = _queries(48, d, seed=43, x=x)
    q[:24] = base.astype(np.float32)
    D, I = ix.search(q, 40)
    Do, Io = o.search(q, 40)
    assert np.array_equal(I, Io) and np.allclose(D, Do, atol=1e-5)
    assert ix.stats()[1] >= 24, "the flooded queries must have failed their completeness proof"
    ix.close()


# ---------------------------------------------------------------------------------------------------------------
# merge kernel on its own, and the one-process row-sharded index (two shards on the one GPU of the test box)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("P", [2, 8])
def test_device_merge_kernel_vs_oracle(P):
    """knnx_merge_topk_device (the step after the all-gather / peer copies): ties across shards, short lists, k = 1..64; k > 64 (the
    front end's num_result_ids = 3000 through a rank-per-GPU ShardedIndex) takes the P-way merge of sorted lists."""
    import torch
    from clip_retrieval_amd.distributed import ShardedIndex
    from oracle.knn_oracle import merge_topk

    rng = np.random.default_rng(P)
    for n, k in [(5, 40), (1, 1), (64, 64), (3, 17), (4, 65), (2, 3000)]:
        D = np.sort(rng.standard_normal((P, n, k)).astype(np.float32), axis=-1)[..., ::-1].copy()
        I = rng.permutation(P * n * k).reshape(P, n, k).astype(np.int64) + (1 << 33)  # ids beyond 32 bits
        if k > 64:  # shard-major ids, as row shards have them (shard p holds the ids [p * n * k, (p + 1) * n * k) of this draw)
            I = (np.arange(P * n * k, dtype=np.int64).reshape(P, n, k) + (1 << 33))
        if P > 1 and k > 64:
            D[1] = D[0]  # (k > 64 merges SORTED lists: a whole list of ties keeps shard 1 sorted) -> id order decides
        elif P > 1 and k > 10:
            D[1, :, 10:] = D[0, :, 10:]  # exact score ties across shards -> id order decides
        if k > 30:
            I[P - 1, 0, 30:] = -1  # a short list
            D[P - 1, 0, 30:] = NEG
            I[0, n - 1, :] = -1  # an empty list
            D[0, n - 1, :] = NEG
        Dg, Ig = torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda()
        Dm, Im = ShardedIndex.merge_device(Dg, Ig, k)
        torch.cuda.synchronize()
        Do, Io = merge_topk(D, I, k)
        assert np.array_equal(Im.cpu().numpy(), Io) and np.array_equal(Dm.cpu().numpy(), Do), f"P={P} n={n} k={k}"


def test_sharded_index_two_shards_on_one_gpu_vs_flat_oracle():
    """ShardedMi355xIndex(devices=[0, 0, 0]): three row shards, scans on per-shard streams, device-to-device gather,
    device merge -- the in-process multi-GPU path of KnnService with every device being GPU 0.  search (k <= 64 and the
    large-k path), search_and_reconstruct, reconstruct and range_search must equal the flat oracle over all rows."""
    from clip_retrieval_amd.knn import ShardedMi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d, n = 768, 30_011
    x = _data(n, d, seed=31)
    x[20_000:20_050] = x[5:55]  # duplicates across shards: ties resolved by global id
    o = FlatIPOracle(d)
    o.add(x)
    ix = ShardedMi355xIndex(d, [0, 0, 0])
    ix.reserve(n)
    for lo in range(0, n, 7_000):  # adds that straddle shard boundaries
        ix.add(x[lo:lo + 7_000])
    assert ix.ntotal == n and ix.nshards == 3
    for nq, k in [(1, 40), (37, 40), (70, 5), (2, 64)]:
        q = _queries(nq, d, seed=nq * 7 + k, x=x)
        D, I, R = ix.search_and_reconstruct(q, k)
        Do, Io = o.search(q, k)
        _check(D, I, Do, Io, f"sharded nq={nq} k={k}")
        assert np.array_equal(R, x[I].astype(np.float32))
    q = _queries(3, d, seed=5, x=x)
    D, I = ix.search(q, 200)  # k > 64: per-shard threshold descent + P-way merge of sorted lists on the device
    Do, Io = o.search(q, 200)
    _check(D, I, Do, Io, "sharded k=200")
    ids = np.array([0, 10_003, 10_004, n - 1, -1], dtype=np.int64)
    got = ix.reconstruct_batch(ids)
    assert np.array_equal(got[:4], x[ids[:4]].astype(np.float32)) and np.isnan(got[4]).all()
    lims, Dr, Ir = ix.range_search(q, 0.5)
    lo_, Do_, Io_ = o.range_search(q, 0.5)
    assert np.array_equal(lims, lo_) and np.array_equal(Ir, Io_) and np.allclose(Dr, Do_, atol=1e-5)
    ix.close()


def test_sharded_index_rccl_exchange_single_device_communicator(monkeypatch):
    """The RCCL form of the exchange (ncclCommInitAll + ncclAllGather of the per-shard top-k lists inside one process;
    csrc/knnx_sharded.hip, loaded with dlopen): a communicator cannot hold one GPU twice, so on a one-GPU box the path is exercised
    with ONE shard (KNNX_SHARDS_RCCL=1 forces it) -- communicator creation, the grouped all-gather of scores (float32) and ids
    (int64), the merge behind it, teardown -- and must equal the flat oracle; with three shards on one device the library must fall
    back to peer copies by itself.  More than one GPU has never run this path."""
    from clip_retrieval_amd.knn import ShardedMi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d, n = 768, 20_003
    x = _data(n, d, seed=91)
    o = FlatIPOracle(d)
    o.add(x)
    monkeypatch.setenv("KNNX_SHARDS_RCCL", "1")
    ix = ShardedMi355xIndex(d, [0])
    assert ix.exchange == "rccl", "librccl.so could not be loaded / ncclCommInitAll failed on the one device"
    ix.reserve(n)
    ix.add(x)
    for nq, k in [(1, 40), (33, 40), (70, 7)]:
        q = _queries(nq, d, seed=nq + k, x=x)
        D, I = ix.search(q, k)
        Do, Io = o.search(q, k)
        _check(D, I, Do, Io, f"RCCL exchange nq={nq} k={k}")
    ix.close()
    three = ShardedMi355xIndex(d, [0, 0, 0])
    assert three.exchange == "peer-copies"
    three.close()


def test_sharded_index_adopts_ivf_shards():
    """Config-5 layout on one GPU: every shard is an IVF-Flat index over its row range with the SAME centroids
    (replicated coarse quantiser) and global ids; the sharded handle must return what one IVF index over all rows returns."""
    from clip_retrieval_amd.knn import Mi355xIndex, ShardedMi355xIndex, build_ivf_index
    from oracle.knn_oracle import IVFFlatOracle

    d, n, nlist, nprobe = 768, 12_000, 32, 6
    x = _data(n, d, seed=41)
    cent = x[np.random.default_rng(0).choice(n, nlist, replace=False)]
    bounds = [0, 5_000, n]
    shards = [build_ivf_index(x[bounds[g]:bounds[g + 1]], nlist, nprobe=nprobe, id_base=bounds[g], centroids=cent) for g in range(2)]
    lists = np.concatenate([sh.ivf_lists for sh in shards])
    ora = IVFFlatOracle(d, cent, lists, x)
    ix = ShardedMi355xIndex.from_shards(shards, bounds[:2])
    q = _queries(20, d, seed=3, x=x)
    D, I = ix.search(q, 10)
    Do, Io = ora.search(q, 10, nprobe)
    _check(D, I, Do, Io, "sharded ivf")
    # the wrapper reports the nprobe its adopted shards carry (it used to say 1 until set, and a "restore" then wrote 1)
    assert ix.nprobe == nprobe
    ix.nprobe = 9
    assert ix.nprobe == 9 and all(int(ix._lib.knnx_ivf_nprobe(C.c_void_p(ix._lib.knnx_shards_get(ix._h, g)))) == 9 for g in range(2))
    ix.nprobe = nprobe
    ix.close()
    assert all(sh._h is None for sh in shards)  # ownership moved


def test_uncoalesced_dedup_requests_from_many_threads_do_not_share_staging():
    """with coalescing off, knnx_search_dedup reaches the batch runner from every request thread at once; the per-index
    staging vectors are touched under the index mutex now.  32 threads x 6 requests against a coalesce=False index must each get
    exactly what a serial call gets (ids, scores, R, links)."""
    import threading

    from clip_retrieval_amd.knn import Mi355xIndex

    d, n = 768, 40_000
    x = _data(n, d, seed=77)
    x[1000:1010] = x[0:10]  # near-duplicates: some requests have links
    ix = Mi355xIndex(d, coalesce=False)
    ix.add(x)
    qs = _queries(32 * 6, d, seed=78, x=x)
    want = [ix.search_dedup(qs[i:i + 1], 40, 0.94, want_r=True) for i in range(len(qs))]
    got = [None] * len(qs)
    errs = []

    def work(t):
        try:
            for j in range(6):
                i = t * 6 + j
                got[i] = ix.search_dedup(qs[i:i + 1], 40, 0.94, want_r=True)
        except Exception as e:  # pylint: disable=broad-except
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(32)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[:3]
    for i, (w, g) in enumerate(zip(want, got)):
        assert np.array_equal(w[1], g[1]) and np.array_equal(w[0], g[0]) and np.array_equal(w[2], g[2]), f"request {i} differs from its serial answer"
        assert (w[3] is None and g[3] is None) or np.array_equal(w[3], g[3]), f"request {i}: links differ"
    ix.close()


def test_load_index_from_numpy_writer_output(tmp_path):
    """`clip inference` output -> index (takes the place of clip_back.py:589-596): the img_emb_*.npy files NumpyWriter
    writes (writer.py:67-75) are loaded in partition order, ids = global row order; row_range loads one shard with
    id_base = lo; devices=[...] builds the in-process sharded object.  All three must answer like the oracle."""
    from clip_retrieval_amd.knn import load_index
    from clip_retrieval_amd.writer import NumpyWriter
    from oracle.knn_oracle import FlatIPOracle

    d = 512
    parts = [_data(m, d, seed=50 + i) for i, m in enumerate((700, 1, 1300))]
    for i, p in enumerate(parts):
        w = NumpyWriter(partition_id=i, output_folder=str(tmp_path), enable_text=False, enable_image=True,
                        enable_metadata=False, output_partition_count=3)
        for lo in range(0, len(p), 512):  # batches as the Runner delivers them
            w({"image_embs": p[lo:lo + 512], "text_embs": None, "image_filename": [f"{i}_{j}" for j in range(lo, min(lo + 512, len(p)))],
               "text": None, "metadata": None})
        w.flush()
    x = np.concatenate(parts)
    o = FlatIPOracle(d)
    o.add(x)
    q = _queries(9, d, seed=1, x=x)
    Do, Io = o.search(q, 40)
    folder = str(tmp_path / "img_emb")
    ix = load_index(folder)
    assert ix.ntotal == len(x) and ix.d == d
    D, I = ix.search(q, 40)
    _check(D, I, Do, Io, "load_index")
    ix.close()
    sh = load_index(folder, devices=[0, 0])
    D, I, R = sh.search_and_reconstruct(q, 40)
    _check(D, I, Do, Io, "load_index devices=[0,0]")
    assert np.array_equal(R, x[I].astype(np.float32))
    sh.close()
    lo, hi = 650, 1500  # a shard that spans all three files
    part = load_index(folder, row_range=(lo, hi))
    op = FlatIPOracle(d)
    op.add(x[lo:hi])
    Dp, Ip = part.search(q, 40)
    Dpo, Ipo = op.search(q, 40)
    _check(Dp, Ip, Dpo, Ipo + lo, "load_index row_range")
    part.close()


def test_ivf_index_from_a_folder_saved_and_loaded_back(tmp_path):
    """Row f1 finished (VERDICT r3 missing #1; clip_index.py:12-66 builds once, clip_back.py:589-596 / 883-896 boots from the file):
    a `clip inference` output folder (NumpyWriter's img_emb_*.npy, 70 k rows over five partitions incl. an empty one) is
    streamed into an IVF-Flat index (sample -> k-means, assignment pass, scatter pass), saved (centroids + 4 bytes of list id per
    row + manifest), and `load_index(folder)` re-creates it WITHOUT k-means or assignment: identical id lists and scores before /
    after, equal to the numpy oracle on the same centroids and lists; `row_range=` (one shard) and `devices=[...]` (the in-process
    sharded object) load from the same files; a changed embeddings folder is refused."""
    import json
    import os

    from clip_retrieval_amd import knn
    from clip_retrieval_amd.writer import NumpyWriter
    from oracle.knn_oracle import IVFFlatOracle

    d, nlist, nprobe = 512, 96, 12
    sizes = (30_000, 0, 17_500, 1, 22_500)
    emb = tmp_path / "embeddings"
    parts = []
    rng = np.random.default_rng(9)
    centres = rng.standard_normal((200, d)).astype(np.float32)
    for i, m in enumerate(sizes):
        x = centres[rng.integers(0, 200, m)] + 0.7 * rng.standard_normal((m, d)).astype(np.float32)
        x = (x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)).astype(np.float16)
        parts.append(x)
        w = NumpyWriter(partition_id=i, output_folder=str(emb), enable_text=False, enable_image=True, enable_metadata=False,
                        output_partition_count=len(sizes))
        if m:
            w({"image_embs": x, "text_embs": None, "image_filename": [str(j) for j in range(m)], "text": None, "metadata": None})
        w.flush()
    x = np.concatenate(parts)
    n = len(x)
    folder = str(emb / "img_emb")
    built = knn.build_ivf_index_from_folder(folder, nlist, nprobe=nprobe, niter=4, seed=1, chunk=8192)
    assert built.ntotal == n and built.nlist == nlist and built.nprobe == nprobe
    lists = built.ivf_lists
    assert lists.shape == (n,) and lists.min() >= 0 and lists.max() < nlist and np.bincount(lists, minlength=nlist).min() > 0
    q = _queries(40, d, seed=4, x=x)
    ora = IVFFlatOracle(d, built.ivf_centroids, lists, x)
    Do, Io = ora.search(q, 40, nprobe)
    Db, Ib, Rb = built.search_and_reconstruct(q, 40)
    _check(Db, Ib, Do, Io, "ivf from folder")
    assert np.array_equal(Rb[Ib >= 0], x[Ib[Ib >= 0]].astype(np.float32))
    # recall of the trained lists against the exact answer: the lists are real clusters, not a random partition
    exact = np.argsort(-(q @ x.astype(np.float32).T), axis=1, kind="stable")[:, :10]
    assert np.mean([len(set(exact[i]) & set(Ib[i])) / 10 for i in range(len(q))]) > 0.8

    out = str(tmp_path / "indices" / "image.index")
    man = knn.save_index(built, out)
    assert sorted(os.listdir(out)) == ["ivf_centroids.npy", "ivf_lists.npy", "ivf_manifest.json"]
    assert man["row_range"] == [0, n] and man["embeddings"]["files"] == [[f"img_emb_{i}.npy", m] for i, m in enumerate(sizes) if m]  # (an empty partition writes no file: writer.py:58-60)
    assert os.path.getsize(os.path.join(out, "ivf_lists.npy")) == 128 + 4 * n  # 4 bytes per row
    built.close()

    loaded = knn.load_index(out)  # no training, no assignment: one scatter pass
    assert loaded.ntotal == n and loaded.nlist == nlist and loaded.nprobe == nprobe
    Dl, Il, Rl = loaded.search_and_reconstruct(q, 40)
    assert np.array_equal(Il, Ib) and np.array_equal(Dl, Db) and np.array_equal(Rl, Rb), "the loaded index answers differently"
    loaded.nprobe = nlist  # every list: the flat answer
    D, I = loaded.search(q[:8], 10)
    assert np.array_equal(I, exact[:8])
    loaded.close()

    lo, hi = 25_000, 52_000  # a shard spanning three partition files; ids stay global
    shard = knn.load_index(out, row_range=(lo, hi))
    Ds, Is = shard.search(q, 40)
    Dso, Iso = IVFFlatOracle(d, np.load(os.path.join(out, "ivf_centroids.npy")), lists[lo:hi], x[lo:hi]).search(q, 40, nprobe)
    _check(Ds, Is, Dso, np.where(Iso >= 0, Iso + lo, -1), "ivf shard from a saved index")
    shard.close()

    sharded = knn.load_index(out, devices=[0, 0, 0])  # three shards of one saved index (all on this GPU), merged top-k
    assert sharded.ntotal == n and sharded.nprobe == nprobe
    Dm, Im, Rm = sharded.search_and_reconstruct(q, 40)
    _check(Dm, Im, Do, Io, "ivf sharded from a saved index")
    assert np.array_equal(Rm[Im >= 0], x[Im[Im >= 0]].astype(np.float32))
    sharded.nprobe = nlist
    D, I = sharded.search(q[:8], 10)
    assert np.array_equal(I, exact[:8])
    sharded.close()

    # a second shard trained on ITS OWN rows would have other centroids: shards share one set through `centroids=`
    other = knn.build_ivf_index_from_folder(folder, nlist, nprobe=nprobe, row_range=(lo, hi), centroids=np.load(os.path.join(out, "ivf_centroids.npy")), chunk=8192)
    assert np.array_equal(other.ivf_lists, lists[lo:hi])
    man2 = knn.save_index(other, str(tmp_path / "indices" / "shard1"))
    assert man2["row_range"] == [lo, hi]
    other.close()
    with pytest.raises(ValueError):
        knn.load_index(str(tmp_path / "indices" / "shard1"), row_range=(0, 10))  # outside the saved shard

    # an index built from an in-memory array (build_ivf_index) is saved by naming the folder its rows can be re-read from
    mem = knn.build_ivf_index(x, nlist, nprobe=nprobe, centroids=np.load(os.path.join(out, "ivf_centroids.npy")))
    with pytest.raises(ValueError):
        knn.save_index(mem, str(tmp_path / "indices" / "from_memory"))
    knn.save_index(mem, str(tmp_path / "indices" / "from_memory"), embeddings_folder=folder)
    Dm2, Im2 = mem.search(q, 40)
    mem.close()
    again = knn.load_index(str(tmp_path / "indices" / "from_memory"))
    Da, Ia = again.search(q, 40)
    assert np.array_equal(Ia, Im2) and np.array_equal(Da, Dm2) and np.array_equal(Ia, Ib)
    again.close()

    # ids are row numbers: an index must not be loaded over embeddings that changed
    np.save(os.path.join(folder, "img_emb_3.npy"), x[:2])
    with pytest.raises(ValueError):
        knn.load_index(out)
    with open(os.path.join(out, "ivf_manifest.json"), encoding="utf-8") as f:
        assert json.load(f)["format"] == knn.IVF_FORMAT


# ---------------------------------------------------------------------------------------------------------------
# IVF build on the device (assignment kernel, Lloyd update, streaming scatter) and nprobe > 64
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,nlist,n", [(768, 100, 5000), (512, 1000, 70_001), (1024, 33, 777), (256, 4096, 3000), (1024, 130, 5000),
                                       (1024, 4096, 3000), (768, 16384, 2000)])
def test_ivf_assignment_kernel_vs_numpy_argmax(d, nlist, n):
    """knn_assign_kernel (every workgroup streams all centroids, a lane keeps the running best of its point): list ids must
    equal numpy's argmax of the fp32 scores of the same fp16 data, except where the two best scores are within 1e-5."""
    from clip_retrieval_amd.knn import IvfBuilder

    x = _data(n, d, seed=61)
    cent = _data(nlist, d, seed=62)
    cent[nlist // 2] = cent[nlist // 3]  # an exact duplicate centroid: ties go to the smaller id
    b = IvfBuilder(d, nlist)
    b.set_centroids(cent)
    got = b.assign(x)
    b.close()
    s = x.astype(np.float32) @ cent.astype(np.float32).T
    want = s.argmax(1)
    bad = np.flatnonzero(got != want)
    top2 = np.sort(s[bad], axis=1)[:, -2:]
    assert got.min() >= 0 and got.max() < nlist
    assert (np.abs(top2[:, 1] - s[bad, got[bad]]) < 1e-5).all(), f"{bad.size} assignments differ beyond near-ties"
    assert bad.size <= n // 200
    dup = got == nlist // 2
    assert not dup.any(), "a tie between identical centroids must go to the smaller id"


def test_ivf_device_kmeans_improves_and_build_is_consistent():
    """Lloyd iterations on the device (assign + per-list mean): the k-means objective (mean best score) must not decrease,
    every list of the built index holds exactly the rows assigned to it, and search over all lists equals the flat oracle."""
    from clip_retrieval_amd.knn import IvfBuilder, build_ivf_index, train_ivf_centroids
    from oracle.knn_oracle import FlatIPOracle

    d, n, nlist = 768, 30_000, 64
    rng = np.random.default_rng(7)
    centers = rng.standard_normal((nlist, d)).astype(np.float32)
    x = centers[rng.integers(0, nlist, n)] + 0.7 * rng.standard_normal((n, d)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
    objs = []
    for it in (1, 4):
        c = train_ivf_centroids(x, nlist, niter=it, seed=0)
        objs.append(float((x.astype(np.float32) @ c.astype(np.float32).T).max(1).mean()))
    assert objs[1] >= objs[0] - 1e-4 and objs[1] > 0.5, objs
    ix = build_ivf_index(x, nlist, nprobe=nlist, centroids=c)
    assert ix.ntotal == n and ix.nlist == nlist
    b = IvfBuilder(d, nlist)
    b.set_centroids(c)
    assert np.array_equal(b.assign(x), ix.ivf_lists)
    b.close()
    o = FlatIPOracle(d)
    o.add(x)
    q = _queries(20, d, seed=2, x=x)
    D, I = ix.search(q, 10)  # nprobe = nlist: every list is scanned -> the flat answer
    Do, Io = o.search(q, 10)
    _check(D, I, Do, Io, "ivf all lists")
    assert np.array_equal(ix.reconstruct_batch(np.array([0, 12_345, n - 1])), x[[0, 12_345, n - 1]].astype(np.float32))
    ix.close()


@pytest.mark.parametrize("nprobe", [65, 100, 256, 300])
def test_ivf_nprobe_above_64(nprobe):
    """BASELINE config 5 asks for nprobe up to 256: the coarse quantiser dumps all centroid scores and a selection kernel
    marks the nprobe best lists (score desc, list id asc)."""
    from clip_retrieval_amd.knn import build_ivf_index
    from oracle.knn_oracle import IVFFlatOracle

    d, n, nlist = 512, 20_000, 300
    x = _data(n, d, seed=71)
    cent = x[np.random.default_rng(1).choice(n, nlist, replace=False)]
    ix = build_ivf_index(x, nlist, nprobe=min(nprobe, nlist), centroids=cent)
    ora = IVFFlatOracle(d, cent, ix.ivf_lists, x)
    for nq in (37, 64, 256):  # (more than 32: the multi-block pass, whose coarse scan dumps the scores of every block at once)
        q = _queries(nq, d, seed=5 + nq, x=x)
        D, I = ix.search(q, 40)
        Do, Io = ora.search(q, 40, min(nprobe, nlist))
        _check(D, I, Do, Io, f"ivf nprobe={nprobe} nq={nq}")
    ix.close()


@pytest.mark.parametrize("nprobe", [4, 70])
def test_ivf_multi_block_pass_equals_the_32_query_passes(nprobe, monkeypatch):
    """Round 6: a call of up to 256 queries is ONE pass -- every 32-query block runs the same exact scan side by side
    in one launch.  Its D / I must be those of the 32-queries-per-pass path (KNNX_IVF_MULTI=0, read at index creation) bit for bit,
    and the tiles it reports: sum over the blocks >= union over all queries >= the largest single block."""
    from clip_retrieval_amd.knn import build_ivf_index

    d, n, nlist = 768, 40_000, 200
    x = _data(n, d, seed=13)
    cent = x[np.random.default_rng(2).choice(n, nlist, replace=False)]
    q = _queries(231, d, seed=99, x=x)
    ix = build_ivf_index(x, nlist, nprobe=nprobe, centroids=cent)
    ix.profile(True)
    D, I = ix.search(q, 40)
    tiles, union = ix.last_scan_tiles(), ix.last_scan_union_tiles()
    ix.profile(False)
    ix.profile_get()
    per_block = []
    for o in range(0, 231, 32):
        ix.search(q[o:o + 32], 40)
        per_block.append(ix.last_scan_tiles())
    assert tiles == sum(per_block), (tiles, per_block)
    assert max(per_block) <= union <= tiles
    ix.close()
    monkeypatch.setenv("KNNX_IVF_MULTI", "0")
    ix0 = build_ivf_index(x, nlist, nprobe=nprobe, centroids=cent)
    D0, I0 = ix0.search(q, 40)
    ix0.close()
    assert np.array_equal(I, I0) and np.array_equal(D.view(np.uint32), D0.view(np.uint32))
    # (32 queries and fewer are one block of the same pass: its coarse quantiser is the score dump + radix select, the old path's the
    # top-nprobe queue scan)
    ix = build_ivf_index(x, nlist, nprobe=nprobe, centroids=cent)
    for nq in (1, 20, 32):
        D1, I1 = ix.search(q[:nq], 40)
        assert np.array_equal(I1, I0[:nq]) and np.array_equal(D1.view(np.uint32), D0[:nq].view(np.uint32)), nq
    ix.close()


# ------------------------------------------------------------------------------------------------------------
# round 3: device-resident IVF build (BASELINE config 5), the mixture corpus, long range lists, build validation
# ------------------------------------------------------------------------------------------------------------
def _torch_alloc(nbytes):
    import torch

    t = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda:0")
    return t.data_ptr(), t


@pytest.mark.parametrize("d,n_clusters", [(1024, 64), (256, 7), (768, 5000)])
def test_mixture_corpus_is_bit_identical_to_the_cpu_derivation(d, n_clusters):
    """knnx_synth_rows_device kind 1 (config 5's overlapping mixture) against oracle.knn_oracle.synth_mixture_rows, with a
    row offset and a stride; kind 0 against synth_rows through the same entry point."""
    import torch

    from clip_retrieval_amd.knn import synth_rows_device
    from oracle.knn_oracle import synth_mixture_rows, synth_rows

    n, row0, stride, seed = 1500, 123_456_789, 977, 5
    buf = torch.empty((n, d), dtype=torch.float16, device="cuda:0")
    synth_rows_device(buf.data_ptr(), row0, n, d, seed, kind=1, n_clusters=n_clusters, row_stride=stride)
    got = buf.cpu().numpy()
    want = synth_mixture_rows(row0 + stride * np.arange(n), d, seed, n_clusters)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), f"{(got != want).sum()} halves differ"
    nrm = np.linalg.norm(got.astype(np.float32), axis=1)
    assert np.abs(nrm - 1).max() < 2e-3
    synth_rows_device(buf.data_ptr(), 1000, n, d, 3, kind=0)
    assert np.array_equal(buf.cpu().numpy().view(np.uint16), synth_rows(1000 + np.arange(n), d, 3).view(np.uint16))


def test_ivf_build_from_device_rows_equals_the_host_build():
    """build_ivf_index_device (rows produced on the GPU, two passes, positions assigned in row order) must lay out exactly
    the index build_ivf_index makes from the same rows and centroids: same lists, same search results as the IVF oracle,
    same reconstruct; k-means on the device-resident sample must not lose to its seeding."""
    import torch

    from clip_retrieval_amd.knn import IvfBuilder, build_ivf_index, build_ivf_index_device, synth_rows_device
    from oracle.knn_oracle import IVFFlatOracle, synth_mixture_rows

    d, n, nlist, seed, ncl = 1024, 40_000, 128, 5, 16

    def fill(dst, row0, count, stride):
        synth_rows_device(dst, row0, count, d, seed, kind=1, n_clusters=ncl, row_stride=stride)

    ix, st = build_ivf_index_device(fill, n, d, nlist, nprobe=8, niter=4, seed=1, chunk=16_384, alloc=_torch_alloc, keep_lists=True,
                                    points_per_centroid=64)
    assert ix.ntotal == n and ix.nlist == nlist and st["n_sample"] == 8192
    assert int(st["list_sizes"].sum()) == n and np.array_equal(np.bincount(ix.ivf_lists, minlength=nlist), st["list_sizes"])
    x = synth_mixture_rows(np.arange(n), d, seed, ncl)
    # the centroids the device training arrived at are not exposed by the index: train again (deterministic: same sample,
    # same seed, fixed summation order) and assign through the host-pointer entry point of the same kernel
    cent = None
    b = IvfBuilder(d, nlist)
    try:
        from clip_retrieval_amd.knn import train_ivf_centroids_device

        sample = torch.from_numpy(x[:: n // 8192][:8192].copy()).to("cuda:0")
        train_ivf_centroids_device(b, sample.data_ptr(), 8192, niter=4, seed=1)
        cent = b.centroids()
        lists_host = b.assign(x)
    finally:
        b.close()
    assert np.array_equal(lists_host, ix.ivf_lists), "device pass 1 and the host-pointer assignment disagree"
    # spherical k-means: unit-norm centroids, so the mean best score is comparable with that of unit-norm seed points
    assert np.abs(np.linalg.norm(cent.astype(np.float32), axis=1) - 1).max() < 2e-3
    obj_seed = float((x.astype(np.float32) @ x[:: n // 8192][:nlist].astype(np.float32).T).max(1).mean())
    obj = float((x.astype(np.float32) @ cent.astype(np.float32).T).max(1).mean())
    assert obj > obj_seed, (obj, obj_seed)
    sizes = np.bincount(ix.ivf_lists, minlength=nlist)
    assert np.median(sizes) > 0.3 * n / nlist, f"unbalanced lists: min / median / max = {sizes.min()} / {np.median(sizes)} / {sizes.max()}"
    ref = build_ivf_index(x, nlist, nprobe=8, centroids=cent)
    ora = IVFFlatOracle(d, cent, ix.ivf_lists, x)
    q = _queries(33, d, seed=9, x=x)
    for npb in (1, 8, 128):
        ix.nprobe = npb
        ref.nprobe = npb
        D, I = ix.search(q, 40)
        Dr, Ir = ref.search(q, 40)
        Do, Io = ora.search(q, 40, npb)
        _check(D, I, Do, Io, f"device-built ivf nprobe={npb}")
        assert np.array_equal(I, Ir) and np.array_equal(D, Dr), "device-built and host-built index answer differently"
    ids = np.array([0, 1, 17_123, n - 1])
    assert np.array_equal(ix.reconstruct_batch(ids), x[ids].astype(np.float32))
    ix.close()
    ref.close()


def test_ivf_add_assigned_refuses_bad_slots():
    """a position past its list or a (list, position) used twice must be refused, not scattered over a neighbour."""
    from clip_retrieval_amd._lib import HipLibraryError, check
    from clip_retrieval_amd.knn import Mi355xIndex

    d, nlist = 256, 4
    cent = _data(nlist, d, 1)
    rows = _data(6, d, 2)
    sizes = np.array([2, 1, 0, 3], dtype=np.int64)

    def begin():
        ix = Mi355xIndex(d)
        check(ix._lib, ix._lib.knnx_ivf_begin(ix._h, nlist, cent.ctypes.data, sizes.ctypes.data), "knnx")
        return ix

    def add(ix, r, ids, lists, pos):
        r = np.ascontiguousarray(r)
        ids, lists, pos = np.asarray(ids, np.int64), np.asarray(lists, np.int32), np.asarray(pos, np.int32)
        return ix._lib.knnx_ivf_add_assigned(ix._h, r.ctypes.data, len(ids), ids.ctypes.data, lists.ctypes.data, pos.ctypes.data)

    ix = begin()
    assert add(ix, rows[:1], [0], [1], [1]) == -1, "position 1 of a 1-row list"
    assert add(ix, rows[:1], [0], [2], [0]) == -1, "a row for an empty list"
    assert add(ix, rows[:2], [0, 1], [0, 0], [1, 1]) == -1, "the same slot twice in one call"
    assert add(ix, rows[:2], [0, 1], [0, 0], [0, 1]) == 0
    assert add(ix, rows[2:3], [2], [0], [1]) == -1, "a slot that an earlier call filled"
    assert add(ix, rows[2:6], [2, 3, 4, 5], [1, 3, 3, 3], [0, 2, 0, 1]) == 0
    check(ix._lib, ix._lib.knnx_ivf_end(ix._h), "knnx")
    ix.nprobe = nlist
    D, I = ix.search(rows.astype(np.float32), 1)
    assert np.array_equal(I[:, 0], np.arange(6)), "every row finds itself after the refused calls"
    ix.close()
    ix = begin()
    assert add(ix, rows[:2], [0, 1], [0, 0], [0, 1]) == 0
    with pytest.raises(HipLibraryError):
        check(ix._lib, ix._lib.knnx_ivf_end(ix._h), "knnx")  # fewer rows than announced
    ix.close()


def test_range_search_long_hit_lists_go_through_the_radix_sort():
    """> 4096 hits per query: ids must still leave ascending and complete (flat and IVF), ADVICE r2."""
    from clip_retrieval_amd.knn import Mi355xIndex, build_ivf_index
    from oracle.knn_oracle import FlatIPOracle, IVFFlatOracle

    d, n = 256, 70_000
    x = _data(n, d, 5)
    q = _queries(3, d, 6, x)
    ix, o = Mi355xIndex(d, id_base=1_000_000), FlatIPOracle(d)
    ix.add(x)
    o.add(x)
    for thr in (0.0, -0.05, 0.12):
        lims, D, I = ix.range_search(q, thr)
        lo, Do, Io = o.range_search(q, thr)
        s = o.scores(q)
        for i in range(q.shape[0]):
            got, want = I[lims[i]:lims[i + 1]] - 1_000_000, Io[lo[i]:lo[i + 1]]
            assert (np.diff(got) > 0).all(), "ids ascending"
            for r in set(got.tolist()) ^ set(want.tolist()):
                assert abs(s[i, r] - thr) < 2e-6
            both = np.intersect1d(got, want)
            assert np.allclose(D[lims[i]:lims[i + 1]][np.isin(got, both)], s[i, both], atol=1e-5), "scores travel with their ids"
        assert max(np.diff(lims)) > 4096 or thr > 0.1
    ix.close()
    nlist = 16
    cent = x[:nlist]
    iv = build_ivf_index(x, nlist, nprobe=9, centroids=cent)
    ora = IVFFlatOracle(d, cent, iv.ivf_lists, x)
    lims, D, I = iv.range_search(q, -0.02)
    lo, Do, Io = ora.range_search(q, -0.02, 9)
    assert max(np.diff(lims)) > 4096
    for i in range(q.shape[0]):
        got, want = I[lims[i]:lims[i + 1]], Io[lo[i]:lo[i + 1]]
        assert (np.diff(got) > 0).all()
        assert len(set(got.tolist()) ^ set(want.tolist())) <= 2
    iv.close()


def test_large_k_with_a_tight_cluster_of_near_duplicates():
    """100 near-duplicates of the query and nothing else close -- the threshold descent stalls at 100 < k hits for
    several scans; a flat index must keep lowering the threshold (and answer exactly) instead of fetching the whole index."""
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    d, n, k = 256, 50_000, 300
    rng = np.random.default_rng(3)
    x = _data(n, d, 8).astype(np.float32)
    base = x[7].copy()
    dup = rng.choice(n, 100, replace=False)
    x[dup] = base + 1e-3 * rng.standard_normal((100, d)).astype(np.float32) / np.sqrt(d)
    x = x.astype(np.float16)
    ix, o = Mi355xIndex(d), FlatIPOracle(d)
    ix.add(x)
    o.add(x)
    q = base.reshape(1, -1).astype(np.float32)
    D, I = ix.search(q, k)
    Do, Io = o.search(q, k)
    # the 100 near-duplicates score within ~1e-6 of each other: their relative ORDER is fp32-summation noise (sets and scores
    # are still held to the usual bar)
    _check(D, I, Do, Io, "tight cluster, k=300", min_exact=0.6)
    ix.close()
