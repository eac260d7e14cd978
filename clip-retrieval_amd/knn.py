"""faiss-`Index`-shaped object over the MI355X kNN library (search half of the hot path).

Mirrors exactly the part of the faiss Index API the reference exercises:
  * `search_and_reconstruct(x, k) -> (D, I, R)`   clip_retrieval/clip_back.py:362
  * `search(x, k) -> (D, I)`                      clip_retrieval/clip_filter.py:55
  * `range_search(x, thresh) -> (lims, D, I)`     clip_retrieval/clip_filter.py:52, clip_back.py:294
  * `.ntotal`, `.d`, `add(x)`, `reconstruct(i)`   ivf_metadata_ordering.py:51, clip_back.py:292-293
so an instance can be dropped into `ClipResource.image_index` / `.text_index` (clip_back.py:781-782).
`load_index` builds one from the `img_emb/*.npy` folder `clip inference` writes (writer.py:67-75),
taking the place of clip_back.py:589-596 for our index type.

Same argument meaning and error behaviour as faiss: float32 C-contiguous [n, d] queries (anything else
raises like faiss' SWIG wrapper does), int64 labels, -1 / -FLT_MAX padding, exceptions on misuse.
Thread-safe: clip_back serves each request on its own werkzeug thread with n=1 (clip_back.py:1018);
concurrent callers are coalesced into one HBM scan of up to 256 queries (_SCAN_QUERIES) by a leader/follower batcher.
"""

import ctypes as C
import glob
import os
import threading

import numpy as np

from ._lib import HipLibraryError, check, load_library

METRIC_INNER_PRODUCT = 0
_SCAN_QUERIES = 256  # queries one HBM scan serves at most (KNN_RQ_MAX in csrc/knn_kernels.h: the RQ scan; 64: the wide scan)


def _as_queries(x, d):
    x = np.asarray(x)
    if x.dtype != np.float32:
        raise TypeError(f"queries must be float32 (got {x.dtype}); faiss raises the same way")
    if x.ndim != 2 or x.shape[1] != d:
        raise AssertionError(f"queries must have shape [n, {d}], got {x.shape}")
    return np.ascontiguousarray(x)


class _FaissShaped:
    """The part of the faiss Index surface clip_back / clip_filter exercise, on top of `_search_raw`,
    `reconstruct_batch`, `range_search` and `ntotal` of the concrete index (one GPU or row-sharded)."""

    def _init_queue(self, coalesce):
        self._coalesce = coalesce
        self._q_lock = threading.Lock()
        self._pending = []  # [(k, want_r, query_row, slot)]
        self._leader_active = False

    def _search_coalesced(self, q, k, want_r):
        """n == 1 callers from many threads: the first becomes the leader and serves every query queued
        while the GPU was busy, up to one scan's worth, in a single pass over HBM."""
        slot = {"ev": threading.Event(), "out": None, "err": None}
        with self._q_lock:
            self._pending.append((k, want_r, q[0], slot))
            lead = not self._leader_active
            if lead:
                self._leader_active = True
        if not lead:
            slot["ev"].wait()
        else:
            while True:
                with self._q_lock:
                    if not self._pending:
                        self._leader_active = False
                        break
                    k0, r0 = self._pending[0][0], self._pending[0][1]
                    take = [p for p in self._pending if p[0] == k0 and p[1] == r0][:_SCAN_QUERIES]
                    taken = set(id(p) for p in take)
                    self._pending = [p for p in self._pending if id(p) not in taken]
                try:
                    qq = np.stack([p[2] for p in take]).astype(np.float32)
                    D, I, R = self._search_raw(qq, k0, r0)
                    for j, p in enumerate(take):
                        p[3]["out"] = (D[j:j + 1], I[j:j + 1], R[j:j + 1] if R is not None else None)
                except Exception as e:  # pylint: disable=broad-except
                    for p in take:
                        p[3]["err"] = e
                for p in take:
                    p[3]["ev"].set()
        if slot["err"] is not None:
            raise slot["err"]
        return slot["out"]

    def _do_search(self, x, k, want_r):
        if k <= 0:
            raise AssertionError("k must be positive")
        q = _as_queries(x, self.d)
        if q.shape[0] == 0:
            return (np.empty((0, k), np.float32), np.empty((0, k), np.int64),
                    np.empty((0, k, self.d), np.float32) if want_r else None)
        if self._coalesce and q.shape[0] == 1:
            return self._search_coalesced(q, int(k), want_r)
        return self._search_raw(q, int(k), want_r)

    def search(self, x, k):
        D, I, _ = self._do_search(x, k, False)
        return D, I

    def search_and_reconstruct(self, x, k):
        return self._do_search(x, k, True)

    def reconstruct(self, key):
        return self.reconstruct_batch(np.asarray([key], dtype=np.int64))[0]

    def _pad(self, x):
        if self._dpad == self.d:
            return x
        out = np.zeros((x.shape[0], self._dpad), dtype=x.dtype)
        out[:, : self.d] = x
        return out

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass


class Mi355xIndex(_FaissShaped):
    """Flat inner-product index with fp16 rows resident in HBM (one GPU)."""

    def __init__(self, d, device=0, id_base=0, coalesce=True):
        self._lib = load_library()
        self.d = int(d)
        self._dpad = (self.d + 255) // 256 * 256  # kernels want d % 256 == 0; extra columns are zeros
        self.device = int(device)
        h = C.c_void_p()
        check(self._lib, self._lib.knnx_create(self.device, self._dpad, METRIC_INNER_PRODUCT, C.byref(h)), "knnx")
        self._h = h
        self.metric_type = METRIC_INNER_PRODUCT
        self.is_trained = True
        if id_base:
            check(self._lib, self._lib.knnx_set_id_base(self._h, int(id_base)), "knnx")
        # concurrent single-query callers are coalesced INSIDE the library (knnx.h: knnx_set_coalesce; round 4 -- round 3 did it
        # here with a Python condition variable, and the GIL hand-offs capped a served index at 6.5 k requests/s)
        self._init_queue(False)
        check(self._lib, self._lib.knnx_set_coalesce(self._h, 1 if coalesce else 0), "knnx")

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.knnx_destroy(h)

    @property
    def ntotal(self):
        return int(self._lib.knnx_ntotal(self._h))

    # ------------------------------------------------------------------ building
    def reserve(self, n_rows):
        check(self._lib, self._lib.knnx_reserve(self._h, int(n_rows)), "knnx")

    def add(self, x):
        """faiss Index.add: float32 (rounded to fp16 on the device) or float16 rows."""
        x = np.asarray(x)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise AssertionError(f"add expects [n, {self.d}], got {x.shape}")
        if x.dtype == np.float16:
            x = np.ascontiguousarray(self._pad(x))
            check(self._lib, self._lib.knnx_add_f16(self._h, x.ctypes.data, x.shape[0]), "knnx")
        elif x.dtype == np.float32:
            x = np.ascontiguousarray(self._pad(x))
            check(self._lib, self._lib.knnx_add_f32(self._h, x.ctypes.data, x.shape[0]), "knnx")
        else:
            raise TypeError(f"add expects float32 or float16 rows, got {x.dtype}")

    def reset(self):
        """faiss Index.reset(): drop the rows, keep the HBM arena."""
        check(self._lib, self._lib.knnx_reset(self._h), "knnx")

    def attach_device_rows(self, dev_ptr, n_rows):
        """Borrow fp16 rows already in HBM (e.g. a torch tensor's data_ptr()); caller keeps them alive."""
        if self._dpad != self.d:
            raise HipLibraryError("attached rows must already be padded to a multiple of 256 columns")
        check(self._lib, self._lib.knnx_attach_device_f16(self._h, C.c_void_p(int(dev_ptr)), int(n_rows)), "knnx")

    def synth_fill(self, n_rows, seed):
        """Fill the index with the benchmark's synthetic corpus (oracle/knn_oracle.py:synth_rows)."""
        if self._dpad != self.d:
            raise HipLibraryError("synthetic fill needs d % 256 == 0")
        check(self._lib, self._lib.knnx_synth_fill(self._h, int(n_rows), C.c_uint64(int(seed))), "knnx")

    # ------------------------------------------------------------------ IVF-Flat
    def set_ivf_lists(self, centroids, list_sizes, ids):
        """Turn the index into a faiss-IndexIVFFlat-shaped one.  The rows must have been add()ed grouped by list;
        `centroids` [nlist, d] (stored fp16), `list_sizes` [nlist], `ids` [ntotal] = the id of every added row."""
        c = np.ascontiguousarray(self._pad(np.asarray(centroids).astype(np.float16)))
        sizes = np.ascontiguousarray(list_sizes, dtype=np.int64)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        if c.shape[0] != sizes.shape[0] or ids.shape[0] != self.ntotal:
            raise AssertionError("centroids/list_sizes/ids disagree with the index")
        check(self._lib, self._lib.knnx_ivf_set_lists(self._h, c.shape[0], c.ctypes.data, sizes.ctypes.data, ids.ctypes.data), "knnx")

    @property
    def nlist(self):
        return int(self._lib.knnx_ivf_nlist(self._h))

    @property
    def nprobe(self):
        return int(self._lib.knnx_ivf_nprobe(self._h)) if self.nlist > 0 else getattr(self, "_nprobe", 1)

    @nprobe.setter
    def nprobe(self, v):
        """faiss `extract_index_ivf(index).nprobe = v` (clip_back.py:357-369)."""
        check(self._lib, self._lib.knnx_ivf_set_nprobe(self._h, int(v)), "knnx")
        self._nprobe = int(v)

    def last_scan_tiles(self):
        """IVF: 32-row tiles walked by the most recent scan (bytes read = tiles * 32 * d_padded * 2)."""
        t = C.c_int64(0)
        check(self._lib, self._lib.knnx_ivf_last_scan_tiles(self._h, C.byref(t)), "knnx")
        return int(t.value)

    def last_scan_union_tiles(self):
        """IVF: tiles of the union of the lists the most recent (multi-block) pass probed, each list counted once (profiling on)."""
        t = C.c_int64(0)
        check(self._lib, self._lib.knnx_ivf_last_scan_union_tiles(self._h, C.byref(t)), "knnx")
        return int(t.value)

    # ------------------------------------------------------------------ searching
    def _search_raw(self, q, k, want_r):
        n = q.shape[0]
        qp = np.ascontiguousarray(self._pad(q))
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        R = np.empty((n, k, self._dpad), dtype=np.float32) if want_r else None
        check(self._lib, self._lib.knnx_search(self._h, qp.ctypes.data, n, int(k), D.ctypes.data, I.ctypes.data,
                                               R.ctypes.data if want_r else None), "knnx")
        if want_r and self._dpad != self.d:
            R = np.ascontiguousarray(R[:, :, : self.d])
        return D, I, R

    def reconstruct_batch(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.empty((keys.shape[0], self._dpad), dtype=np.float32)
        check(self._lib, self._lib.knnx_reconstruct(self._h, keys.ctypes.data, keys.shape[0], out.ctypes.data), "knnx")
        return np.ascontiguousarray(out[:, : self.d])

    DEDUP_PAIR_CAP = 512

    def search_dedup(self, x, k, threshold=0.94, want_r=False):
        """One request of KnnService.knn_search with its dedup fused (knnx.h: knnx_search_dedup; clip_back.py:362 + :290-309):
        x float32 [1, d], k <= 64 -> (D [1, k], I [1, k], R [1, k, d] or None, links int32 [n_links, 2] of result ranks i < j
        whose normalised stored vectors have inner product > threshold, or None when there were more links than the device
        keeps -- take the general path then).  Concurrent callers share one scan, one gather and one link launch."""
        q = np.ascontiguousarray(self._pad(_as_queries(x, self.d)))
        if q.shape[0] != 1 or not 0 < k <= 64:
            raise AssertionError("search_dedup serves one query and k <= 64")
        k = int(k)
        D = np.empty((1, k), dtype=np.float32)
        I = np.empty((1, k), dtype=np.int64)
        R = np.empty((1, k, self._dpad), dtype=np.float32) if want_r else None
        pairs = np.empty((self.DEDUP_PAIR_CAP, 2), dtype=np.int32)
        n = C.c_int(0)
        check(self._lib, self._lib.knnx_search_dedup(self._h, q.ctypes.data, k, D.ctypes.data, I.ctypes.data, R.ctypes.data if want_r else None,
                                                     C.c_float(threshold), pairs.ctypes.data, self.DEDUP_PAIR_CAP, C.byref(n)), "knnx")
        if want_r and self._dpad != self.d:
            R = np.ascontiguousarray(R[:, :, : self.d])
        return D, I, R, (pairs[: n.value].copy() if n.value <= self.DEDUP_PAIR_CAP else None)

    def i8_served(self):
        """Queries answered through the int8 first stage of the flat scans (include/knnx.h: knnx_i8_served); results are exact
        either way."""
        return int(self._lib.knnx_i8_served(self._h))

    def i8_rows(self):
        """Rows the int8 copy holds at the moment (include/knnx.h: knnx_i8_rows): ntotal, fewer for a partial copy, 0 for none."""
        return int(self._lib.knnx_i8_rows(self._h))

    def i8_planes(self):
        """0: no int8 copy at the moment; 1 / 2: int8 planes per query of the first stage (include/knnx.h: knnx_i8_planes)."""
        return int(self._lib.knnx_i8_planes(self._h))

    def i8_dominant(self):
        """Columns the int8 first stage treats as dominant (include/knnx.h: knnx_i8_dominant): a list of 0 .. 4 column numbers."""
        cols = (C.c_int * 4)()
        n = int(self._lib.knnx_i8_dominant(self._h, cols))
        return [int(cols[j]) for j in range(max(n, 0))]

    def coalesce_stats(self):
        """(batches served, queries in them, largest batch) of the library's request coalescer."""
        b, q, m = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self._lib, self._lib.knnx_coalesce_stats(self._h, C.byref(b), C.byref(q), C.byref(m)), "knnx")
        return int(b.value), int(q.value), int(m.value)

    def range_search(self, x, thresh):
        q = np.ascontiguousarray(self._pad(_as_queries(x, self.d)))
        n = q.shape[0]
        lims = np.zeros(n + 1, dtype=np.int64)
        # one pass when the hits fit a guessed capacity (the per-request dedup finds ~n hits for n vectors); the two-call
        # protocol (count, then fill) otherwise
        guess = max(4096, 16 * n)
        D = np.empty(guess, dtype=np.float32)
        I = np.empty(guess, dtype=np.int64)
        rc = self._lib.knnx_range_search_once(self._h, q.ctypes.data, n, C.c_float(thresh), lims.ctypes.data, D.ctypes.data, I.ctypes.data, guess)
        if rc == 0:
            total = int(lims[n])
            return lims, D[:total].copy(), I[:total].copy()
        if rc < 0:
            check(self._lib, rc, "knnx")
        total = int(lims[n])
        D = np.empty(total, dtype=np.float32)
        I = np.empty(total, dtype=np.int64)
        if total:
            check(self._lib, self._lib.knnx_range_search(self._h, q.ctypes.data, n, C.c_float(thresh), lims.ctypes.data,
                                                         D.ctypes.data, I.ctypes.data), "knnx")
        return lims, D, I

    # ------------------------------------------------------------------ device-buffer path (bench, sharded search)
    def search_device(self, q_ptr, n, k, D_ptr, I_ptr, stream=None):
        check(self._lib, self._lib.knnx_search_device(self._h, C.c_void_p(int(q_ptr)), int(n), int(k), C.c_void_p(int(D_ptr)),
                                                      C.c_void_p(int(I_ptr)), C.c_void_p(int(stream)) if stream else None), "knnx")

    def profile(self, on):
        check(self._lib, self._lib.knnx_profile_enable(self._h, 1 if on else 0), "knnx")

    def stats(self):
        """(queries served by a proof-based scan, queries whose exactness proof failed and were re-run exactly)."""
        a, b = C.c_int64(0), C.c_int64(0)
        check(self._lib, self._lib.knnx_get_stats(self._h, C.byref(a), C.byref(b)), "knnx")
        return int(a.value), int(b.value)

    def profile_get(self):
        n, ms = C.c_int64(0), C.c_double(0.0)
        check(self._lib, self._lib.knnx_profile_get(self._h, C.byref(n), C.byref(ms)), "knnx")
        return int(n.value), float(ms.value)


class ShardedMi355xIndex(_FaissShaped):
    """Row-sharded index over several GPUs of ONE process behind the same faiss-shaped surface -- the object that goes
    into `ClipResource.image_index` (clip_back.py:781-782) when the index needs more than one GPU (BASELINE config 5).
    Shard g = rows [g*T/G, (g+1)*T/G) on devices[g]; ids = global row numbers.  C ABI: knnx_shards_* (include/knnx.h)."""

    def __init__(self, d, devices, coalesce=True, _adopt=None):
        self._lib = load_library()
        self.d = int(d)
        self._dpad = (self.d + 255) // 256 * 256
        self.devices = [int(x) for x in devices]
        h = C.c_void_p()
        dv = (C.c_int * len(self.devices))(*self.devices)
        if _adopt is None:
            check(self._lib, self._lib.knnx_shards_create(len(self.devices), dv, self._dpad, METRIC_INNER_PRODUCT, C.byref(h)), "knnx")
        else:
            shards, row_lo = _adopt
            hs = (C.c_void_p * len(shards))(*[sh._h for sh in shards])  # pylint: disable=protected-access
            lo = (C.c_int64 * len(shards))(*[int(x) for x in row_lo])
            check(self._lib, self._lib.knnx_shards_adopt(len(shards), hs, dv, lo, C.byref(h)), "knnx")
            for sh in shards:
                sh._h = None  # pylint: disable=protected-access  (owned by the sharded handle now)
        self._h = h
        self.metric_type = METRIC_INNER_PRODUCT
        self.is_trained = True
        self._init_queue(coalesce)

    @classmethod
    def from_shards(cls, shards, row_lo, coalesce=True):
        """Adopt per-device `Mi355xIndex` objects (flat or IVF-Flat; shard g built with id_base = row_lo[g])."""
        return cls(shards[0].d, [sh.device for sh in shards], coalesce=coalesce, _adopt=(shards, row_lo))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.knnx_shards_destroy(h)

    @property
    def ntotal(self):
        return int(self._lib.knnx_shards_ntotal(self._h))

    @property
    def nlist(self):
        """Lists of the (shared) coarse quantiser when the shards are IVF-Flat, else 0."""
        return int(self._lib.knnx_ivf_nlist(C.c_void_p(self._lib.knnx_shards_get(self._h, 0)))) if self.nshards else 0

    @property
    def nprobe(self):
        """What the shards carry (knnx_ivf_nprobe of the first IVF shard): an index adopted with from_shards() keeps the value its
        shards were built with (the wrapper used to answer 1 until it had been set, and knn_search's wide path then
        'restored' 1 on every shard)."""
        for g in range(self.nshards):
            sh = C.c_void_p(self._lib.knnx_shards_get(self._h, g))
            if self._lib.knnx_ivf_nlist(sh) > 0:
                return int(self._lib.knnx_ivf_nprobe(sh))
        return getattr(self, "_nprobe", 1)

    @nprobe.setter
    def nprobe(self, v):
        """IVF-Flat shards: faiss `extract_index_ivf(index).nprobe = v` on every shard (clip_back.py:357-369)."""
        for g in range(self.nshards):
            sh = C.c_void_p(self._lib.knnx_shards_get(self._h, g))
            if self._lib.knnx_ivf_nlist(sh) > 0:
                check(self._lib, self._lib.knnx_ivf_set_nprobe(sh, int(v)), "knnx")
        self._nprobe = int(v)

    @property
    def nshards(self):
        return int(self._lib.knnx_shards_count(self._h))

    @property
    def exchange(self):
        """'rccl' (one grouped ncclAllGather of the per-shard top-k lists) or 'peer-copies' (include/knnx.h: knnx_shards_exchange)."""
        return "rccl" if int(self._lib.knnx_shards_exchange(self._h)) == 1 else "peer-copies"

    def reserve(self, total_rows):
        """Fix the row range of every shard; must precede add()."""
        check(self._lib, self._lib.knnx_shards_reserve(self._h, int(total_rows)), "knnx")

    def add(self, x):
        x = np.asarray(x)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise AssertionError(f"add expects [n, {self.d}], got {x.shape}")
        if x.dtype == np.float16:
            x = np.ascontiguousarray(self._pad(x))
            check(self._lib, self._lib.knnx_shards_add_f16(self._h, x.ctypes.data, x.shape[0]), "knnx")
        elif x.dtype == np.float32:
            x = np.ascontiguousarray(self._pad(x))
            check(self._lib, self._lib.knnx_shards_add_f32(self._h, x.ctypes.data, x.shape[0]), "knnx")
        else:
            raise TypeError(f"add expects float32 or float16 rows, got {x.dtype}")

    def synth_fill(self, rows_per_shard, seed):
        if self._dpad != self.d:
            raise HipLibraryError("synthetic fill needs d % 256 == 0")
        check(self._lib, self._lib.knnx_shards_synth_fill(self._h, int(rows_per_shard), C.c_uint64(int(seed))), "knnx")

    def _search_raw(self, q, k, want_r):
        n = q.shape[0]
        qp = np.ascontiguousarray(self._pad(q))
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        R = np.empty((n, k, self._dpad), dtype=np.float32) if want_r else None
        check(self._lib, self._lib.knnx_shards_search(self._h, qp.ctypes.data, n, int(k), D.ctypes.data, I.ctypes.data,
                                                      R.ctypes.data if want_r else None), "knnx")
        if want_r and self._dpad != self.d:
            R = np.ascontiguousarray(R[:, :, : self.d])
        return D, I, R

    def reconstruct_batch(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.empty((keys.shape[0], self._dpad), dtype=np.float32)
        check(self._lib, self._lib.knnx_shards_reconstruct(self._h, keys.ctypes.data, keys.shape[0], out.ctypes.data), "knnx")
        return np.ascontiguousarray(out[:, : self.d])

    def range_search(self, x, thresh):
        q = np.ascontiguousarray(self._pad(_as_queries(x, self.d)))
        n = q.shape[0]
        lims = np.zeros(n + 1, dtype=np.int64)
        check(self._lib, self._lib.knnx_shards_range_search(self._h, q.ctypes.data, n, C.c_float(thresh), lims.ctypes.data, None, None), "knnx")
        total = int(lims[n])
        D = np.empty(total, dtype=np.float32)
        I = np.empty(total, dtype=np.int64)
        if total:
            check(self._lib, self._lib.knnx_shards_range_search(self._h, q.ctypes.data, n, C.c_float(thresh), lims.ctypes.data,
                                                                D.ctypes.data, I.ctypes.data), "knnx")
        return lims, D, I

    def shard_stats(self):
        """Per-shard (proof-served queries, proof failures)."""
        out = []
        for g in range(self.nshards):
            a, b = C.c_int64(0), C.c_int64(0)
            check(self._lib, self._lib.knnx_get_stats(C.c_void_p(self._lib.knnx_shards_get(self._h, g)), C.byref(a), C.byref(b)), "knnx")
            out.append((int(a.value), int(b.value)))
        return out


def embedding_files(folder):
    """The `img_emb_*.npy` / `text_emb_*.npy` files of one `clip inference` output folder, in partition
    order (zero-padded names sort correctly: writer.py:22,67)."""
    # (not the sidecars save_index() writes -- ivf_centroids.npy / ivf_lists.npy: an index saved INTO its embeddings folder must not
    # turn into two more partitions at the next load)
    files = sorted(f for f in glob.glob(os.path.join(folder, "*.npy")) if not os.path.basename(f).startswith("ivf_"))
    if not files:
        raise ValueError(f"no .npy embedding files under {folder}")
    return files


def load_index(path, device=0, row_range=None, enable_faiss_memory_mapping=False, devices=None):  # pylint: disable=unused-argument
    """Build an HBM-resident flat index from a folder of fp16 `.npy` partitions -- or, when `path` is a folder written by
    `save_index()` (it holds ivf_manifest.json), re-create that IVF-Flat index without training or assigning anything.

    Takes the place of clip_back.py:589-596 (`faiss.read_index`) for this index type; ids are the global
    row order of the concatenated partitions = the metadata row order (clip_back.py:401-417).
    `row_range=(lo, hi)` loads one shard of a row-sharded index and sets its id base to `lo` (one process per GPU);
    `devices=[0, 1, ...]` builds ONE object that row-shards the whole folder over those GPUs of this process
    (`ShardedMi355xIndex`, the KnnService case).  `enable_faiss_memory_mapping` is accepted for call compatibility:
    rows are always resident in HBM; the files themselves are read through np.load(mmap_mode="r").
    """
    if os.path.isfile(os.path.join(path, IVF_MANIFEST)):  # a built IVF-Flat index saved by save_index(): no k-means, no assignment
        return _load_ivf_index(path, device=device, row_range=row_range, devices=devices)
    src = FolderRows(path)
    files, shapes, d, total = src.files, src.shapes, src.d, src.n
    lo, hi = (0, total) if row_range is None else row_range
    if devices is not None:
        if row_range is not None:
            raise ValueError("row_range and devices are mutually exclusive")
        index = ShardedMi355xIndex(d, devices)
        index.reserve(total)
    else:
        index = Mi355xIndex(d, device=device, id_base=lo)
        index.reserve(max(hi - lo, 0))
    start = 0
    for f, s in zip(files, shapes):
        a0, a1 = max(lo, start), min(hi, start + s[0])
        if a1 > a0:
            a = np.load(f, mmap_mode="r")[a0 - start:a1 - start]
            index.add(np.ascontiguousarray(a) if a.dtype in (np.float16, np.float32) else np.asarray(a, dtype=np.float32))
        start += s[0]
    return index


# ------------------------------------------------------------------------------------------------------------
# IVF-Flat index build (BASELINE config 5; takes the place of the autofaiss call of clip_index.py:12-66 for this
# index type).  The arithmetic runs on the GPU through the builder entry points of include/knnx.h: list assignment is
# the MFMA assignment kernel (csrc/knn_rq_kernels.hip: knn_assign_kernel), the Lloyd update one workgroup per list, the
# final layout a scatter kernel; the host keeps integer bookkeeping only (bincount, prefix sums, stable ranks).
# ------------------------------------------------------------------------------------------------------------
class IvfBuilder:
    """Centroids (and optionally a training sample) resident on one GPU: knnx_ivfb_* of include/knnx.h."""

    def __init__(self, d, nlist, device=0):
        self._lib = load_library()
        self.d, self.nlist, self.device = int(d), int(nlist), int(device)
        self._dpad = (self.d + 255) // 256 * 256
        h = C.c_void_p()
        check(self._lib, self._lib.knnx_ivfb_create(self.device, self._dpad, self.nlist, C.byref(h)), "knnx")
        self._h = h

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.knnx_ivfb_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass

    def _rows(self, x):
        x = np.asarray(x)
        if x.dtype != np.float16:
            x = x.astype(np.float16)
        if self._dpad != self.d:
            out = np.zeros((x.shape[0], self._dpad), dtype=np.float16)
            out[:, : self.d] = x
            x = out
        return np.ascontiguousarray(x)

    def set_centroids(self, c):
        c = self._rows(c)
        assert c.shape[0] == self.nlist
        check(self._lib, self._lib.knnx_ivfb_set_centroids(self._h, c.ctypes.data), "knnx")

    def centroids(self):
        out = np.empty((self.nlist, self._dpad), dtype=np.float16)
        check(self._lib, self._lib.knnx_ivfb_get_centroids(self._h, out.ctypes.data), "knnx")
        return np.ascontiguousarray(out[:, : self.d])

    def set_sample(self, x):
        x = self._rows(x)
        self._n_sample = x.shape[0]
        check(self._lib, self._lib.knnx_ivfb_set_sample(self._h, x.ctypes.data, x.shape[0]), "knnx")

    def assign_sample(self):
        out = np.empty(self._n_sample, dtype=np.int32)
        check(self._lib, self._lib.knnx_ivfb_assign_sample(self._h, out.ctypes.data), "knnx")
        return out

    def update(self, lists):
        """One Lloyd update from the resident sample; returns the list sizes (empty lists keep their centroid)."""
        order = np.ascontiguousarray(np.argsort(lists, kind="stable").astype(np.int64))
        sizes = np.bincount(lists, minlength=self.nlist).astype(np.int64)
        off = np.zeros(self.nlist + 1, dtype=np.int64)
        np.cumsum(sizes, out=off[1:])
        check(self._lib, self._lib.knnx_ivfb_update(self._h, order.ctypes.data, off.ctypes.data), "knnx")
        return sizes

    def assign(self, x):
        """List id (argmax over the centroids, ties -> smaller id) of every row of x; rows are streamed in 1 Mi-row chunks."""
        x = self._rows(x)
        out = np.empty(x.shape[0], dtype=np.int32)
        check(self._lib, self._lib.knnx_ivfb_assign(self._h, x.ctypes.data, x.shape[0], out.ctypes.data), "knnx")
        return out


    # ---- rows that are already in HBM (device pointers; SURVEY 8 row f1 at BASELINE config 5's size) ----
    def set_sample_device(self, dev_ptr, n):
        """Borrow n fp16 rows [n, d_padded] of device memory as the training sample (the caller keeps them alive)."""
        self._n_sample = int(n)
        check(self._lib, self._lib.knnx_ivfb_set_sample_device(self._h, C.c_void_p(dev_ptr), int(n)), "knnx")

    def seed_from_sample(self, list_ids, sample_rows):
        """centroid list_ids[i] := sample row sample_rows[i] (initial seeding / re-seeding of empty lists)."""
        li = np.ascontiguousarray(list_ids, dtype=np.int32)
        sr = np.ascontiguousarray(sample_rows, dtype=np.int64)
        assert li.shape == sr.shape
        check(self._lib, self._lib.knnx_ivfb_seed_from_sample(self._h, li.ctypes.data, sr.ctypes.data, li.size), "knnx")

    def lloyd(self):
        """One Lloyd iteration over the resident sample, driven inside the library; returns the list sizes."""
        sizes = np.empty(self.nlist, dtype=np.int64)
        check(self._lib, self._lib.knnx_ivfb_lloyd(self._h, sizes.ctypes.data), "knnx")
        return sizes

    def assign_device(self, rows_ptr, n, lists_ptr):
        """lists[i] = list of device row i (int32 device array); the list sizes accumulate in the builder."""
        check(self._lib, self._lib.knnx_ivfb_assign_device(self._h, C.c_void_p(rows_ptr), int(n), C.c_void_p(lists_ptr)), "knnx")

    def list_sizes(self, reset=False):
        sizes = np.empty(self.nlist, dtype=np.int64)
        check(self._lib, self._lib.knnx_ivfb_list_sizes(self._h, sizes.ctypes.data, int(bool(reset))), "knnx")
        return sizes


def synth_rows_device(dst_ptr, row_begin, n, d, seed, kind=0, n_clusters=0, row_stride=1, device=0, stream=None):
    """Benchmark corpus rows generated into device memory (knnx_synth_rows_device): kind 0 isotropic, 1 config-5 mixture, 2 isotropic
    with three dominant columns (CLIP-like anisotropy)."""
    lib = load_library()
    check(lib, lib.knnx_synth_rows_device(int(device), C.c_void_p(dst_ptr), int(row_begin), int(row_stride), int(n), int(d),
                                          C.c_uint64(seed), int(kind), int(n_clusters), C.c_void_p(stream) if stream else None), "knnx")


def rebalance_centroids(cent, sizes, rng, small=0.5, big=2.0, eps=0.05):
    """Split-and-merge step of the k-means (what faiss' Clustering does for empty clusters -- `split_clusters` -- extended to the
    small ones): every list with fewer than `small` x the mean size (smallest first, and only while it is under a quarter of the list it
    would split) gives its centroid up, and the currently largest list (above `big` x the mean) is split by replacing its centroid c with the unit vectors of c + delta and c - delta (delta = eps |c| times a
    random sign vector / sqrt(d)): the next Lloyd iteration deals the big list's members between the two halves, the small list's
    few members move to their next-nearest centroids.  Round 5: BASELINE config 5's shard had lists of 1 / 1 880 /
    14 993 rows (min / median / max) and scanned 1.2 - 1.3 x the bytes of the balanced-list model.  Returns the number of splits;
    `cent` (float32 [nlist, d]) is changed in place."""
    import heapq  # pylint: disable=import-outside-toplevel

    sizes = np.asarray(sizes, dtype=np.float64)
    mean = sizes.mean()
    donors = [int(i) for i in np.argsort(sizes) if sizes[i] < small * mean]
    heap = [(-float(sizes[i]), int(i)) for i in np.flatnonzero(sizes > big * mean)]
    heapq.heapify(heap)
    taken = set(donors)
    n = 0
    for i in donors:
        while heap and heap[0][1] in taken:
            heapq.heappop(heap)
        if not heap or -heap[0][0] <= big * mean:
            break
        if sizes[i] > -heap[0][0] / 4:  # giving up a list a quarter the size of the one it would split gains nothing
            break
        s, j = heapq.heappop(heap)
        c = cent[j].astype(np.float64)
        delta = eps * np.linalg.norm(c) / np.sqrt(c.size) * rng.choice((-1.0, 1.0), size=c.size)
        a, b = c + delta, c - delta
        cent[j] = (a / max(np.linalg.norm(a), 1e-30)).astype(cent.dtype)
        cent[i] = (b / max(np.linalg.norm(b), 1e-30)).astype(cent.dtype)
        heapq.heappush(heap, (s / 2, j))  # each half may be split again
        heapq.heappush(heap, (s / 2, i))
        taken.discard(i)
        n += 1
    return n


def train_ivf_centroids_device(builder, sample_ptr, n_sample, niter=8, seed=0, balance=True):
    """k-means over a device-resident sample with `builder` (IvfBuilder): seeds from distinct random sample rows, runs
    `niter` Lloyd iterations, re-seeds empty lists on random sample rows and -- `balance`, all but the last two iterations -- moves the
    centroids of the smallest lists into the largest ones (rebalance_centroids).  Returns the list sizes of the last iteration."""
    rng = np.random.default_rng(seed)
    builder.set_sample_device(sample_ptr, n_sample)
    builder.seed_from_sample(np.arange(builder.nlist), np.sort(rng.choice(n_sample, builder.nlist, replace=False)))
    sizes = None
    for it in range(niter):
        sizes = builder.lloyd()
        if balance and it < niter - 2 and builder.nlist >= 16:
            cent = builder.centroids().astype(np.float32)
            if rebalance_centroids(cent, sizes, rng):
                builder.set_centroids(cent.astype(np.float16))
                continue
        empty = np.flatnonzero(sizes == 0)
        if empty.size:
            builder.seed_from_sample(empty, rng.choice(n_sample, empty.size, replace=False))
    return sizes


def _download_i32(dev_ptr, n, device):  # pylint: disable=unused-argument
    """n int32 from device memory to a numpy array (plain hipMemcpy through the HIP runtime libclipx.so already links)."""
    out = np.empty(int(n), dtype=np.int32)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    rc = hip.hipMemcpy(out.ctypes.data, C.c_void_p(int(dev_ptr)), out.nbytes, 2)  # hipMemcpyDeviceToHost
    if rc != 0:
        raise HipLibraryError(f"hipMemcpy (device -> host) failed with code {rc}")
    return out


def _release_cached_device_memory():
    import sys

    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        torch.cuda.empty_cache()


def build_ivf_index_device(fill_rows, n, d, nlist, nprobe=16, niter=8, seed=0, device=0, id_base=0, sample_rows=None,
                           chunk=1 << 20, alloc=None, points_per_centroid=64, keep_lists=False):
    """IVF-Flat index over n rows that are PRODUCED ON THE GPU (BASELINE config 5: one 125 M x 1024 shard = 256 GB of HBM).

    `fill_rows(dst_ptr, row0, count, stride)` writes fp16 rows row0, row0 + stride, ... ([count, d], d a multiple of 256)
    at device address dst_ptr; it is called for the training sample (stride > 1) and twice per chunk (assignment pass and
    scatter pass) -- the corpus never exists twice.  **fill_rows must have COMPLETED when it returns** (synchronise the stream it
    launched on): the library's assignment / scatter kernels read the buffer on the index's own non-blocking streams right after,
    and nothing orders those behind a producer that is still running on another stream.  `alloc(nbytes)` returns (device pointer, keep-alive object) for the
    scratch buffers (sample, one chunk of rows, 4 bytes per row of list ids); default: torch uint8 tensors (torch is
    this package's device-memory plumbing).  Returns (index, stats dict)."""
    import time

    assert d % 256 == 0, "device builds take padded rows (d % 256 == 0)"
    lib = load_library()
    if alloc is None:
        def alloc(nbytes):
            import torch

            t = torch.empty(int(nbytes), dtype=torch.uint8, device=f"cuda:{device}")
            return t.data_ptr(), t
    t0 = time.perf_counter()
    b = IvfBuilder(d, nlist, device)
    n_sample = int(min(n, nlist * points_per_centroid)) if sample_rows is None else int(sample_rows)
    stride = max(1, n // n_sample)
    sample_ptr, sample_keep = alloc(n_sample * d * 2)
    fill_rows(sample_ptr, 0, n_sample, stride)
    train_ivf_centroids_device(b, sample_ptr, n_sample, niter=niter, seed=seed)
    cent = np.ascontiguousarray(b.centroids())
    del sample_keep
    _release_cached_device_memory()  # the arena allocated below needs the bytes a caching allocator would sit on
    t1 = time.perf_counter()
    # pass 1: list of every row (4 bytes per row stay on the device), list sizes
    lists_ptr, lists_keep = alloc(n * 4)
    rows_ptr, rows_keep = alloc(min(chunk, n) * d * 2)
    b.list_sizes(reset=True)
    for o in range(0, n, chunk):
        m = min(chunk, n - o)
        fill_rows(rows_ptr, o, m, 1)
        b.assign_device(rows_ptr, m, lists_ptr + 4 * o)
    sizes = b.list_sizes()
    b.close()
    _release_cached_device_memory()
    t2 = time.perf_counter()
    # pass 2: scatter into the list-sorted arena
    index = Mi355xIndex(d, device=device, id_base=id_base)
    check(lib, lib.knnx_ivf_begin(index._h, nlist, cent.ctypes.data, sizes.ctypes.data), "knnx")  # pylint: disable=protected-access
    for o in range(0, n, chunk):
        m = min(chunk, n - o)
        fill_rows(rows_ptr, o, m, 1)
        check(lib, lib.knnx_ivf_add_assigned_device(index._h, C.c_void_p(rows_ptr), m, id_base + o, C.c_void_p(lists_ptr + 4 * o)), "knnx")  # pylint: disable=protected-access
    check(lib, lib.knnx_ivf_end(index._h), "knnx")  # pylint: disable=protected-access
    if keep_lists:  # tests / recall measurements: the list of every row (4 bytes per row to the host)
        index.ivf_lists = _download_i32(lists_ptr, n, device)
    del rows_keep, lists_keep
    _release_cached_device_memory()
    t3 = time.perf_counter()
    index.nprobe = min(nprobe, nlist)
    stats = {"train_s": t1 - t0, "assign_s": t2 - t1, "scatter_s": t3 - t2, "n_sample": n_sample, "list_sizes": sizes}
    return index, stats


def train_ivf_centroids(x_f16, nlist, niter=8, seed=0, device=0, max_points_per_centroid=256):
    """k-means with inner-product assignment (faiss Clustering with an IndexFlatIP quantiser): returns fp16 [nlist, d].
    The sample (<= nlist * max_points_per_centroid rows, like faiss) stays resident in HBM across the iterations."""
    x_f16 = np.asarray(x_f16)
    rng = np.random.default_rng(seed)
    n, d = x_f16.shape
    if n < nlist:
        raise ValueError(f"need at least nlist={nlist} training rows, got {n}")
    take = min(n, nlist * max_points_per_centroid)
    sample = x_f16[np.sort(rng.choice(n, take, replace=False))] if take < n else x_f16
    b = IvfBuilder(d, nlist, device)
    b.set_sample(sample)
    b.set_centroids(sample[rng.choice(sample.shape[0], nlist, replace=False)])
    for it in range(niter):
        sizes = b.update(b.assign_sample())
        if it < niter - 2 and nlist >= 16:  # split-and-merge: the smallest lists' centroids move into the largest lists
            cent = b.centroids().astype(np.float32)
            if rebalance_centroids(cent, sizes, rng):
                b.set_centroids(cent.astype(np.float16))
                continue
        empty = np.flatnonzero(sizes == 0)
        if empty.size:  # re-seed empty clusters on random points (faiss splits big clusters; any re-seed is valid)
            cent = b.centroids()
            cent[empty] = np.asarray(sample[rng.choice(sample.shape[0], empty.size, replace=False)], dtype=np.float16)
            b.set_centroids(cent)
    cent = b.centroids()
    b.close()
    return cent


def _positions_in_lists(lists, cursor):
    """Stable rank of every row inside its list, continuing from `cursor` (rows of earlier chunks); updates cursor."""
    order = np.argsort(lists, kind="stable")
    sl = lists[order]
    start = np.flatnonzero(np.r_[True, sl[1:] != sl[:-1]])
    run_len = np.diff(np.r_[start, sl.size])
    rank_sorted = np.arange(sl.size) - np.repeat(start, run_len)
    pos = np.empty(lists.size, dtype=np.int64)
    pos[order] = rank_sorted + cursor[sl]
    np.add.at(cursor, sl[start], run_len)
    return pos.astype(np.int32)


def build_ivf_index(x_f16, nlist, nprobe=16, niter=8, seed=0, device=0, id_base=0, centroids=None, chunk=1 << 20):
    """fp16 rows [N, d] -> HBM-resident IVF-Flat index (ids = id_base + row number, like the flat index).  `x_f16` may be a
    numpy memmap: rows are streamed twice in chunks (assignment, then scatter into the list-sorted arena)."""
    n, d = x_f16.shape
    if centroids is None:
        centroids = train_ivf_centroids(x_f16, nlist, niter=niter, seed=seed, device=device)
    centroids = np.asarray(centroids).astype(np.float16)
    b = IvfBuilder(d, nlist, device)
    b.set_centroids(centroids)
    lists = np.empty(n, dtype=np.int32)
    for o in range(0, n, chunk):
        lists[o:o + chunk] = b.assign(x_f16[o:o + chunk])
    b.close()
    sizes = np.bincount(lists, minlength=nlist).astype(np.int64)
    index = Mi355xIndex(d, device=device, id_base=id_base)
    lib = index._lib  # pylint: disable=protected-access
    cpad = np.ascontiguousarray(index._pad(centroids))  # pylint: disable=protected-access
    check(lib, lib.knnx_ivf_begin(index._h, nlist, cpad.ctypes.data, sizes.ctypes.data), "knnx")  # pylint: disable=protected-access
    cursor = np.zeros(nlist, dtype=np.int64)
    for o in range(0, n, chunk):
        rows = np.ascontiguousarray(index._pad(np.asarray(x_f16[o:o + chunk], dtype=np.float16)))  # pylint: disable=protected-access
        ls = np.ascontiguousarray(lists[o:o + chunk])
        pos = _positions_in_lists(ls, cursor)
        ids = np.arange(o, o + rows.shape[0], dtype=np.int64) + id_base
        check(lib, lib.knnx_ivf_add_assigned(index._h, rows.ctypes.data, rows.shape[0], ids.ctypes.data, ls.ctypes.data, pos.ctypes.data), "knnx")  # pylint: disable=protected-access
    check(lib, lib.knnx_ivf_end(index._h), "knnx")  # pylint: disable=protected-access
    index.nprobe = min(nprobe, nlist)
    index.ivf_lists = lists  # kept for tests / recall measurement, and for save_index(index, folder, embeddings_folder=...)
    index.ivf_centroids, index.ivf_row_range = centroids, (int(id_base), int(id_base) + int(n))
    return index


# ------------------------------------------------------------------------------------------------------------
# IVF-Flat from disk and back ("img_emb_*.npy -> resident shards"; row a17: the reference builds once,
# offline -- clip_index.py:12-66 -- and boots by reading the index file -- clip_back.py:589-596, 883-896)
# ------------------------------------------------------------------------------------------------------------
IVF_MANIFEST = "ivf_manifest.json"
IVF_FORMAT = "clip-retrieval_amd ivf-flat v1"


class FolderRows:
    """The rows of one `clip inference` output folder (`img_emb_{zfill}.npy`, writer.py:67-75: NPY v1, C order, fp16 [n_i, d]) as
    ONE virtual [n, d] matrix in partition order -- global row number = id = metadata row (clip_back.py:401-417) -- read through
    np.load(mmap_mode="r"), never resident as a whole."""

    def __init__(self, folder):
        self.folder = folder
        self.files = embedding_files(folder)
        self.shapes = []
        for f in self.files:
            a = np.load(f, mmap_mode="r")
            if a.ndim != 2:
                raise ValueError(f"{f}: expected a 2-D embedding matrix")
            self.shapes.append(tuple(int(v) for v in a.shape))
        self.d = self.shapes[0][1]
        if any(sh[1] != self.d for sh in self.shapes):
            raise ValueError("embedding files disagree on the dimension")
        self.starts = np.concatenate([[0], np.cumsum([sh[0] for sh in self.shapes])]).astype(np.int64)
        self.n = int(self.starts[-1])

    def rows(self, lo, hi):
        """fp16 [hi - lo, d] (a copy), possibly spanning several files."""
        lo, hi = int(lo), int(hi)
        if not 0 <= lo <= hi <= self.n:
            raise IndexError(f"rows [{lo}, {hi}) outside [0, {self.n})")
        parts = []
        f0 = int(np.searchsorted(self.starts, lo, side="right")) - 1
        for fi in range(max(f0, 0), len(self.files)):
            a0, a1 = max(lo, int(self.starts[fi])), min(hi, int(self.starts[fi + 1]))
            if a1 > a0:
                a = np.load(self.files[fi], mmap_mode="r")[a0 - int(self.starts[fi]):a1 - int(self.starts[fi])]
                parts.append(np.asarray(a, dtype=np.float16))
            if int(self.starts[fi + 1]) >= hi:
                break
        if not parts:
            return np.zeros((0, self.d), dtype=np.float16)
        return np.ascontiguousarray(parts[0]) if len(parts) == 1 else np.concatenate(parts)

    def chunks(self, lo, hi, chunk):
        for o in range(int(lo), int(hi), int(chunk)):
            yield o, self.rows(o, min(o + chunk, hi))

    def take(self, idx):
        """fp16 rows at sorted global row numbers `idx` (the k-means training sample)."""
        idx = np.asarray(idx, dtype=np.int64)
        out = np.empty((idx.size, self.d), dtype=np.float16)
        fi = np.searchsorted(self.starts, idx, side="right") - 1
        for f in np.unique(fi):
            sel = np.flatnonzero(fi == f)
            out[sel] = np.load(self.files[int(f)], mmap_mode="r")[idx[sel] - int(self.starts[int(f)])]
        return out

    def manifest(self):
        return {"folder": os.path.abspath(self.folder), "files": [[os.path.basename(f), sh[0]] for f, sh in zip(self.files, self.shapes)],
                "rows": self.n, "d": self.d}


def _scatter_into_ivf(src, lo, hi, centroids, lists, nprobe, device, chunk):
    """The second pass of an IVF build: rows [lo, hi) of `src` into a list-sorted arena on `device` (ids = global row numbers),
    given every row's list id.  Streams the rows once; no training, no assignment."""
    nlist, d = centroids.shape
    sizes = np.bincount(lists, minlength=nlist).astype(np.int64)
    if sizes.shape[0] != nlist:
        raise ValueError("a list id is outside [0, nlist)")
    index = Mi355xIndex(d, device=device, id_base=lo)
    lib = index._lib  # pylint: disable=protected-access
    cpad = np.ascontiguousarray(index._pad(np.asarray(centroids, dtype=np.float16)))  # pylint: disable=protected-access
    check(lib, lib.knnx_ivf_begin(index._h, nlist, cpad.ctypes.data, sizes.ctypes.data), "knnx")  # pylint: disable=protected-access
    cursor = np.zeros(nlist, dtype=np.int64)
    for o, x in src.chunks(lo, hi, chunk):
        rows = np.ascontiguousarray(index._pad(x))  # pylint: disable=protected-access
        ls = np.ascontiguousarray(lists[o - lo:o - lo + rows.shape[0]], dtype=np.int32)
        pos = _positions_in_lists(ls, cursor)
        ids = np.arange(o, o + rows.shape[0], dtype=np.int64)
        check(lib, lib.knnx_ivf_add_assigned(index._h, rows.ctypes.data, rows.shape[0], ids.ctypes.data, ls.ctypes.data, pos.ctypes.data), "knnx")  # pylint: disable=protected-access
    check(lib, lib.knnx_ivf_end(index._h), "knnx")  # pylint: disable=protected-access
    index.nprobe = min(int(nprobe), nlist)
    index.ivf_lists, index.ivf_centroids, index.ivf_source, index.ivf_row_range = lists, np.asarray(centroids, dtype=np.float16), src, (int(lo), int(hi))
    return index


def build_ivf_index_from_folder(path, nlist, nprobe=16, niter=8, seed=0, device=0, row_range=None, centroids=None,
                                max_points_per_centroid=256, chunk=1 << 20):
    """`clip inference` output folder -> HBM-resident IVF-Flat index, streaming the `.npy` partitions (never the whole shard in
    host memory; what `clip_index` / autofaiss do for the reference's index types, clip_index.py:12-66).
      pass 0  a strided sample of <= nlist * max_points_per_centroid rows -> spherical k-means on the GPU (train_ivf_centroids);
              `centroids=` skips it (all shards of a row-sharded index MUST share one set: train once, pass it to the others)
      pass 1  every row through the MFMA assignment kernel (knnx_ivfb_assign): 4 bytes of list id per row stay on the host
      pass 2  knnx_ivf_begin / add_assigned / end: the rows into the list-sorted, tile-padded arena
    `row_range=(lo, hi)`: one shard of a row-sharded index (ids stay global row numbers).  The result can be `save_index()`ed."""
    src = path if isinstance(path, FolderRows) else FolderRows(path)
    lo, hi = (0, src.n) if row_range is None else (int(row_range[0]), int(row_range[1]))
    if centroids is None:
        take = min(src.n, int(nlist) * int(max_points_per_centroid))  # sample the WHOLE folder, so that every shard trains the same
        idx = np.unique(np.linspace(0, src.n - 1, take).astype(np.int64))
        centroids = train_ivf_centroids(src.take(idx), nlist, niter=niter, seed=seed, device=device, max_points_per_centroid=max_points_per_centroid)
    centroids = np.asarray(centroids).astype(np.float16)
    b = IvfBuilder(src.d, nlist, device)
    b.set_centroids(centroids)
    lists = np.empty(hi - lo, dtype=np.int32)
    for o, x in src.chunks(lo, hi, chunk):
        lists[o - lo:o - lo + x.shape[0]] = b.assign(x)
    b.close()
    return _scatter_into_ivf(src, lo, hi, centroids, lists, nprobe, device, chunk)


def save_index(index, folder, embeddings_folder=None):
    """Write a built IVF-Flat index (build_ivf_index_from_folder, or build_ivf_index given `embeddings_folder`) so that a service
    boots WITHOUT k-means or assignment: `ivf_centroids.npy` (fp16 [nlist, d]), `ivf_lists.npy` (int32 list id of every row of the
    shard: 4 bytes per row) and `ivf_manifest.json` (format, d, nlist, nprobe, row range, and the embeddings folder with the name
    and row count of each partition file -- the rows themselves are not copied: the arena is rebuilt from them by one scatter
    pass).  The counterpart of the reference's `<indices>/image.index` written by clip_index (clip_index.py:12-66), read back by
    `load_index(folder)` in the place of clip_back.py:589-596."""
    import json  # pylint: disable=import-outside-toplevel

    lists, cent = getattr(index, "ivf_lists", None), getattr(index, "ivf_centroids", None)
    src = getattr(index, "ivf_source", None)
    if lists is None or cent is None:
        raise ValueError("save_index takes an IVF-Flat index built by build_ivf_index_from_folder / build_ivf_index")
    if src is None:
        if embeddings_folder is None:
            raise ValueError("this index was built from an in-memory array: name the embeddings folder its rows can be re-read from")
        src = FolderRows(embeddings_folder)
    lo, hi = getattr(index, "ivf_row_range", (0, src.n))
    if hi - lo != lists.shape[0]:
        raise ValueError("list ids and row range disagree")
    os.makedirs(folder, exist_ok=True)
    np.save(os.path.join(folder, "ivf_centroids.npy"), np.asarray(cent, dtype=np.float16))
    np.save(os.path.join(folder, "ivf_lists.npy"), np.asarray(lists, dtype=np.int32))
    man = {"format": IVF_FORMAT, "d": int(index.d), "nlist": int(cent.shape[0]), "nprobe": int(index.nprobe), "row_range": [int(lo), int(hi)],
           "embeddings": src.manifest(), "embeddings_relative": os.path.relpath(os.path.abspath(src.folder), os.path.abspath(folder))}
    tmp = os.path.join(folder, IVF_MANIFEST + ".part")
    with open(tmp, "w", encoding="utf-8") as f:
        json.dump(man, f, indent=1)
    os.replace(tmp, os.path.join(folder, IVF_MANIFEST))  # the manifest appears last: a folder without it is not an index
    return man


def read_ivf_manifest(folder, embeddings_folder=None):
    """(manifest dict, FolderRows of the embeddings it names) -- the embeddings are looked for at the recorded relative path
    first (the index folder and the embeddings usually move together), then at the absolute one; their file names and row
    counts must still be what they were when the index was built (ids are row numbers)."""
    import json  # pylint: disable=import-outside-toplevel

    with open(os.path.join(folder, IVF_MANIFEST), encoding="utf-8") as f:
        man = json.load(f)
    if man.get("format") != IVF_FORMAT:
        raise ValueError(f"{folder}: unknown index format {man.get('format')!r}")
    cands = [embeddings_folder] if embeddings_folder else [os.path.normpath(os.path.join(folder, man["embeddings_relative"])), man["embeddings"]["folder"]]
    src = None
    for c in cands:
        if c and os.path.isdir(c) and glob.glob(os.path.join(c, "*.npy")):
            src = FolderRows(c)
            break
    if src is None:
        raise FileNotFoundError(f"{folder}: the embeddings this index was built from are not at {cands}")
    if src.manifest()["files"] != man["embeddings"]["files"] or src.d != man["d"]:
        raise ValueError(f"{src.folder}: the embedding files changed since the index was built (names / row counts / dimension)")
    return man, src


def _load_ivf_index(folder, device=0, row_range=None, devices=None, embeddings_folder=None, chunk=1 << 20):
    man, src = read_ivf_manifest(folder, embeddings_folder)
    cent = np.load(os.path.join(folder, "ivf_centroids.npy"))
    lists = np.load(os.path.join(folder, "ivf_lists.npy"), mmap_mode="r")
    slo, shi = man["row_range"]
    if cent.shape != (man["nlist"], man["d"]) or lists.shape[0] != shi - slo:
        raise ValueError(f"{folder}: sidecar files disagree with the manifest")
    if devices is not None:
        if row_range is not None:
            raise ValueError("row_range and devices are mutually exclusive")
        G = len(devices)
        cuts = [slo + (shi - slo) * g // G for g in range(G + 1)]
        shards = [_scatter_into_ivf(src, cuts[g], cuts[g + 1], cent, np.asarray(lists[cuts[g] - slo:cuts[g + 1] - slo]), man["nprobe"], devices[g], chunk)
                  for g in range(G)]
        sharded = ShardedMi355xIndex.from_shards(shards, cuts[:-1])
        sharded.nprobe = man["nprobe"]
        return sharded
    lo, hi = (slo, shi) if row_range is None else (int(row_range[0]), int(row_range[1]))
    if not slo <= lo <= hi <= shi:
        raise ValueError(f"row_range {row_range} is outside the saved shard's rows [{slo}, {shi})")
    return _scatter_into_ivf(src, lo, hi, cent, np.asarray(lists[lo - slo:hi - slo]), man["nprobe"], device, chunk)
