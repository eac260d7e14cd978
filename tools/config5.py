#!/usr/bin/env python3
"""BASELINE config 5, one GPU's shard at its stated size: IVF-Flat over 125 M x 1024 fp16 rows (256 GB of the 288 GB),
nlist 65 536, nprobe 16 / 64 / 256, built ON the device and served through the KnnService hot path.

    python tools/config5.py [--rows 125000000 --nlist 65536 --nprobe 16,64,256 --threads 1,8,64 --seconds 3]

What it does (reference call sites clip_back.py:343-369 knn_search, :1018 threaded server):
  1. corpus   the overlapping mixture of Gaussians of knnx_synth_rows_device(kind 1) -- generated chunk by chunk on the GPU,
              never resident twice (the rows are re-derivable from (seed, row): the two build passes regenerate them);
  2. build    knn.build_ivf_index_device: k-means on a strided device sample, assignment pass (knn_assign_kernel, MFMA),
              scatter pass into the list-sorted arena;
  3. truth    nprobe = nlist walks every list = the exact top-40 over the WHOLE shard (stronger than the 10 M-row subsample
              SURVEY names: every row is a candidate);
  4. recall   recall@40 of nprobe 16 / 64 / 256 against it, bytes actually scanned (work-list tiles) vs the model
              (nprobe / nlist) * N * d * 2, scan GB/s from hipEvents around the scan kernel;
  5. served   T client threads issuing n = 1 requests through service.KnnHotPath.knn_search (search_and_reconstruct + the
              reference's result handling; concurrent callers are coalesced into one scan INSIDE the library -- knnx_set_coalesce, round 4 --
              and the +dedup leg runs knnx_search_dedup: the links of every request of a batch in one launch): QPS, p50 / p99.
Prints human-readable lines and ONE final JSON line (prefix "CONFIG5 "); bench.py --ivf embeds that object.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def run(rows=125_000_000, d=1024, nlist=65536, clusters=0, nprobes=(16, 64, 256), threads=(1, 8, 64), seconds=3.0, k=40,
        n_queries=1024, seed=5, niter=8, device=0, chunk=1 << 20, points_per_centroid=64, log=print, dedup_leg=True,
        profile_batches=0):
    import numpy as np
    import torch

    from clip_retrieval_amd.knn import build_ivf_index_device, synth_rows_device
    from clip_retrieval_amd.service import KnnHotPath
    from clip_retrieval_amd.synth import perturbed_queries

    torch.cuda.set_device(device)
    clusters = clusters or max(1, nlist // 8)
    free, total = torch.cuda.mem_get_info(device)
    # arena = rows (+ up to 31 pad rows per list) * d * 2, idmap 8 B + inverse map 4 B + list ids 4 B per row, one chunk of rows
    per_row = d * 2 + 16
    fit = int((free - (6 << 30) - nlist * 32 * d * 2 - chunk * d * 2) // per_row)
    if rows > fit:
        log(f"config5: {rows} rows do not fit the {free / 2**30:.1f} GiB that are free; using {fit}")
        rows = fit
    log(f"config5: rows={rows} d={d} ({rows * d * 2 / 1e9:.1f} GB fp16) nlist={nlist} mixture components={clusters} seed={seed} "
        f"HBM free {free / 2**30:.1f} / {total / 2**30:.1f} GiB")

    def fill(dst, row0, count, stride):
        synth_rows_device(dst, row0, count, d, seed, kind=1, n_clusters=clusters, row_stride=stride, device=device)

    def alloc(nbytes):
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=f"cuda:{device}")
        return t.data_ptr(), t

    t0 = time.perf_counter()
    ix, st = build_ivf_index_device(fill, rows, d, nlist, nprobe=nprobes[0], niter=niter, seed=seed, device=device, chunk=chunk,
                                    alloc=alloc, points_per_centroid=points_per_centroid)
    build_s = time.perf_counter() - t0
    sizes = st["list_sizes"]
    assign_tflops = 2.0 * rows * nlist * d / st["assign_s"] / 1e12
    log(f"build: k-means ({niter} iterations on {st['n_sample']} rows) {st['train_s']:.1f} s, assignment pass {st['assign_s']:.1f} s "
        f"({assign_tflops:.0f} TFLOP/s incl. row generation), scatter pass {st['scatter_s']:.1f} s, total {build_s:.1f} s; "
        f"list sizes min / median / max = {int(sizes.min())} / {int(np.median(sizes))} / {int(sizes.max())}")
    free2, _ = torch.cuda.mem_get_info(device)
    log(f"HBM after build: {(total - free2) / 2**30:.1f} GiB in use")

    # queries: perturbed copies of rows spread over the shard
    qstride = max(1, rows // n_queries)
    qrows = torch.empty((n_queries, d), dtype=torch.float16, device=f"cuda:{device}")
    fill(qrows.data_ptr(), qstride // 2, n_queries, qstride)
    planted = qstride // 2 + qstride * np.arange(n_queries, dtype=np.int64)
    q = perturbed_queries(qrows.float().cpu().numpy(), noise=0.1, seed=4)
    del qrows

    if profile_batches:  # rocprofv3 --kernel-trace --stats runs: only the B = 256 / 64 calls, a few of each per nprobe
        for npb in nprobes:
            ix.nprobe = npb
            for B in (256, 64):
                for _ in range(profile_batches):
                    ix.search(q[:B], k)
        ix.close()
        return {"profile_batches": profile_batches}

    def search_all(B=32):
        D = np.empty((n_queries, k), np.float32)
        I = np.empty((n_queries, k), np.int64)
        tiles, t1 = 0, time.perf_counter()
        for o in range(0, n_queries, B):
            D[o:o + B], I[o:o + B] = ix.search(q[o:o + B], k)
            tiles += ix.last_scan_tiles()
        return D, I, tiles, time.perf_counter() - t1

    # ground truth: every list probed = exact search over the whole shard
    ix.nprobe = nlist
    ix.search(q[:32], k)
    ix.profile(True)
    Dt, It, tiles_t, el = search_all()
    ix.profile(False)
    nl, ms = ix.profile_get()
    exact_gbs = tiles_t * 32 * d * 2 / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    top1 = float((It[:, 0] == planted).mean())
    log(f"exact (nprobe = nlist): {n_queries} queries in {el:.2f} s, {nl} scan launches, {ms / max(nl, 1):.2f} ms per 32-query scan, "
        f"{exact_gbs:.0f} GB/s ({exact_gbs / HBM_PEAK_GBS:.3f} of 8 TB/s); planted row is the top hit for {top1:.4f} of the queries")

    hot = KnnHotPath(dedup_device=device)

    class Resource:  # the ClipResource fields knn_search reads (clip_back.py:770-790)
        image_index = ix
        text_index = ix
        metadata_is_ordered_by_ivf = False
        safety_model = None
        violence_detector = None

    out = {"rows": rows, "d": d, "nlist": nlist, "mixture_components": clusters, "k": k, "queries": n_queries,
           "build_s": round(build_s, 2), "train_s": round(st["train_s"], 2), "assign_s": round(st["assign_s"], 2),
           "scatter_s": round(st["scatter_s"], 2), "assign_tflops": round(assign_tflops, 1),
           "list_size_min_median_max": [int(sizes.min()), int(np.median(sizes)), int(sizes.max())],
           "exact": {"ms_per_32_query_scan": round(ms / max(nl, 1), 3), "GBps": round(exact_gbs, 1),
                     "planted_top1": top1},
           "by_nprobe": []}
    for npb in nprobes:
        ix.nprobe = npb
        ix.search(q[:32], k)
        ix.profile(True)
        D, I, tiles, el = search_all()
        ix.profile(False)
        nl, ms = ix.profile_get()
        rec = float(np.mean([len(set(I[i].tolist()) & set(It[i].tolist())) / k for i in range(n_queries)]))
        top1 = float((I[:, 0] == planted).mean())
        scanned = tiles * 32 * d * 2 / (n_queries / 32)          # bytes per 32-query scan (union of the probed lists)
        model1 = npb / nlist * rows * d * 2                       # SURVEY's per-query model, balanced lists
        gbs = tiles * 32 * d * 2 / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # one query at a time: bytes actually walked per query vs the model
        t1 = time.perf_counter()
        tiles1 = 0
        for i in range(64):
            ix.search(q[i:i + 1], k)
            tiles1 += ix.last_scan_tiles()
        lat1 = (time.perf_counter() - t1) / 64
        row = {"nprobe": npb, "recall_at_40_vs_exact_whole_shard": round(rec, 4), "planted_top1": round(top1, 4),
               "batch32": {"qps": round(n_queries / el, 1), "scan_ms": round(ms / max(nl, 1), 3), "scan_GBps": round(gbs, 1),
                           "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_scan": int(scanned)},
               "single_query": {"ms": round(lat1 * 1e3, 3), "bytes_scanned": int(tiles1 * 32 * d * 2 / 64), "bytes_model": int(model1),
                                "scanned_over_model": round(tiles1 * 32 * d * 2 / 64 / model1, 3)},
               "served": []}
        log(f"nprobe {npb:4d}: recall@{k} {rec:.4f} (planted top-1 {top1:.4f}); B=32 {n_queries / el:9.1f} QPS, scan {ms / max(nl, 1):.3f} ms, "
            f"{gbs:.0f} GB/s; n=1 {lat1 * 1e3:.3f} ms, {tiles1 * 32 * d * 2 / 64 / 1e6:.1f} MB scanned vs model {model1 / 1e6:.1f} MB")
        # one call of 64 / 256 queries (bench.py --ivf rows): ONE multi-block pass each since round 6 (every block of
        # 32 queries runs its exact scan side by side in one launch); bytes walked (sum over the blocks) against the union of the
        # lists of all its queries (each list once) and against the balanced-list model
        for B in (64, 256):
            ix.profile(True)
            Db, Ib = ix.search(q[:B], k)
            tb, ub = ix.last_scan_tiles(), ix.last_scan_union_tiles()
            ix.profile(False)
            nlb, msb = ix.profile_get()
            same = bool(np.array_equal(Ib, I[:B]) and np.array_equal(Db, D[:B]))
            t1 = time.perf_counter()
            for _ in range(5):
                ix.search(q[:B], k)
            elb = (time.perf_counter() - t1) / 5
            row[f"batch{B}"] = {"qps": round(B / elb, 1), "ms_per_batch": round(elb * 1e3, 3), "passes": int(nlb), "scan_ms": round(msb, 3),
                                "scan_GBps": round(tb * 32 * d * 2 / (msb * 1e-3) / 1e9, 1) if msb > 0 else 0.0,
                                "bytes_scanned": int(tb * 32 * d * 2), "bytes_union_of_lists": int(ub * 32 * d * 2),
                                "bytes_model": int(B * model1), "scanned_over_union": round(tb / max(ub, 1), 3),
                                "equals_32_query_passes": same}
            log(f"    B={B}: {B / elb:9.1f} QPS, {elb * 1e3:.3f} ms per batch ({nlb} pass, list scan {msb:.3f} ms = "
                f"{row[f'batch{B}']['scan_GBps']:.0f} GB/s; {tb * 32 * d * 2 / 1e9:.2f} GB walked = {tb / max(ub, 1):.3f} x the union of its lists, "
                f"{tb * 32 * d * 2 / (B * model1):.3f} x the balanced model; equals the 32-query passes: {same})")
        legs = [(T, False) for T in threads]
        if dedup_leg:
            legs.append((threads[-1], True))
        for T, dedup in legs:
            lats = [[] for _ in range(T)]
            stop = time.perf_counter() + seconds
            errors = []

            def client(tid, lats=lats, stop=stop, dedup=dedup, errors=errors):
                j = tid
                try:
                    while time.perf_counter() < stop:
                        a = time.perf_counter()
                        hot.knn_search(q[j % n_queries:j % n_queries + 1], "image", k, Resource, dedup, False, False)
                        lats[tid].append(time.perf_counter() - a)
                        j += T
                except Exception as e:  # pylint: disable=broad-except
                    errors.append(repr(e))

            th = [threading.Thread(target=client, args=(i,)) for i in range(T)]
            t1 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            el2 = time.perf_counter() - t1
            if errors:
                raise RuntimeError(errors[0])
            al = np.sort(np.concatenate([np.asarray(x) for x in lats]))
            srow = {"threads": T, "deduplicate": dedup, "qps": round(len(al) / el2, 1), "p50_ms": round(float(al[len(al) // 2]) * 1e3, 3),
                    "p99_ms": round(float(al[min(len(al) - 1, int(len(al) * 0.99))]) * 1e3, 3), "requests": int(len(al))}
            if hasattr(ix, "coalesce_stats"):
                cb, cq, cm = ix.coalesce_stats()
                srow["coalescer"] = {"batches_total": cb, "queries_total": cq, "largest_batch": cm}
            row["served"].append(srow)
            log(f"    served n=1 x {T:3d} threads{' +dedup' if dedup else '       '}: {srow['qps']:9.1f} QPS, p50 {srow['p50_ms']:.3f} ms, "
                f"p99 {srow['p99_ms']:.3f} ms ({srow['requests']} requests through KnnHotPath.knn_search)")
        out["by_nprobe"].append(row)
    best = max(out["by_nprobe"], key=lambda r: r["batch32"]["scan_GBps"])
    out["roofline"] = {"bound": "hbm", "kernel": "knn_scan_kernel<IVF> (work list of the probed lists)", "achieved": best["batch32"]["scan_GBps"],
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": best["batch32"]["hbm_frac"], "nprobe": best["nprobe"],
                       "algorithmic_bytes_per_launch": best["batch32"]["bytes_per_scan"], "avg_launch_ms": best["batch32"]["scan_ms"],
                       "traffic": None}
    ix.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=125_000_000)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--nlist", type=int, default=65536)
    ap.add_argument("--clusters", type=int, default=0, help="mixture components (0: nlist / 8)")
    ap.add_argument("--nprobe", default="16,64,256")
    ap.add_argument("--threads", default="1,8,64")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--niter", type=int, default=8)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--points-per-centroid", type=int, default=64)
    ap.add_argument("--profile-batches", type=int, default=0, help="build, then ONLY this many B = 256 and B = 64 calls per nprobe (for rocprofv3 --kernel-trace --stats)")
    a = ap.parse_args()
    out = run(rows=a.rows, d=a.d, nlist=a.nlist, clusters=a.clusters, nprobes=tuple(int(x) for x in a.nprobe.split(",")),
              threads=tuple(int(x) for x in a.threads.split(",")), seconds=a.seconds, niter=a.niter, n_queries=a.queries,
              points_per_centroid=a.points_per_centroid, log=lambda m: print(m, flush=True), profile_batches=a.profile_batches)
    print("CONFIG5 " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
