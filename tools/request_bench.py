#!/usr/bin/env python3
"""One /knn-service request, measured stage by stage (README.md:429-437 publishes the reference's stage means:
text clip inference 18.6 ms, image 20.6 ms, knn index 26.7 ms, metadata get 41.5 ms on its CPU box).

    python tools/request_bench.py [--rows 100000000 --meta-rows 10000000 --reps 20]

A request = KnnHotPath.compute_query -> knn_search (search_and_reconstruct + post filter) -> map_to_metadata
(clip_back.py:419-470).  Encoder: ViT-L/14 (random-init weights, synthetic BPE vocabulary: neither the checkpoint nor the
vocabulary file exists offline -- the arithmetic per request is the same); index: flat IP over --rows x 768 fp16 rows resident
in HBM (BASELINE config 3); metadata: an Arrow IPC table of --meta-rows rows (url, caption), memory-mapped, through the
reference's per-id slicing (clip_back.py:608-615 restated) and through service.ArrowMetadataProvider's batched take.
Prints a table like README.md:429-437 and one JSON line (prefix "REQUEST ").
"""
import argparse
import base64
import gzip
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_merges(folder):
    from clip_retrieval_amd.tokenizer import BPE_FILE_NAME, bytes_to_unicode

    b2u = bytes_to_unicode()
    merges, seen = [], set()
    for w in "a photo of the cat dog red blue car house tree sky person walking on beach in city at night".split():
        sym = [b2u[b] for b in w.encode()]
        sym[-1] += "</w>"
        while len(sym) > 1:
            p = (sym[0], sym[1])
            if p not in seen:
                seen.add(p)
                merges.append(p)
            sym = [sym[0] + sym[1]] + sym[2:]
    with gzip.open(os.path.join(folder, BPE_FILE_NAME), "wt", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n")


class ReferenceStyleProvider:
    """clip_back.py:599-615 restated: one 1-row slice per id, concatenated (what the batched take replaces)."""

    def __init__(self, table):
        self.table = table

    def get(self, ids, cols=None):
        import pyarrow as pa

        if cols is None:
            cols = self.table.schema.names
        else:
            cols = list(set(self.table.schema.names) & set(cols))
        t = pa.concat_tables([self.table[i:i + 1] for i in ids])
        return t.select(cols).to_pandas().to_dict("records")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--meta-rows", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--model", default="ViT-L/14")
    a = ap.parse_args()

    import numpy as np
    import pyarrow as pa
    import pyarrow.compute as pc
    import torch
    from PIL import Image
    from types import SimpleNamespace

    from clip_retrieval_amd.encoder import load_clip
    from clip_retrieval_amd.knn import Mi355xIndex
    from clip_retrieval_amd.service import ArrowMetadataProvider, KnnHotPath
    from clip_retrieval_amd.synth import perturbed_queries

    tmp = tempfile.mkdtemp(prefix="reqbench")
    synthetic_merges(tmp)
    model, preprocess, tokenizer = load_clip("random:" + a.model, warmup_batch_size=1, clip_cache_path=tmp)
    d = 768
    ix = Mi355xIndex(d)
    free = torch.cuda.mem_get_info()[0]
    rows = int(min(a.rows, (free - (8 << 30)) // (d * 2)))
    ix.synth_fill(rows, 3)
    print(f"index: {rows} x {d} fp16 = {rows * d * 2 / 1e9:.1f} GB resident; metadata table: {a.meta_rows} rows", flush=True)

    # metadata: Arrow IPC file, memory-mapped like the reference's provider
    ids = pa.array(np.arange(a.meta_rows, dtype=np.int64))
    s = pc.cast(ids, pa.string())
    table = pa.table({"url": pc.binary_join_element_wise(pa.scalar("https://example.org/images/"), s, pa.scalar(".jpg"), pa.scalar("")),
                      "caption": pc.binary_join_element_wise(pa.scalar("a synthetic caption for image number "), s, pa.scalar("")),
                      "width": pa.array(np.full(a.meta_rows, 512, dtype=np.int32))})
    mdir = os.path.join(tmp, "meta")
    os.makedirs(mdir)
    with pa.OSFile(os.path.join(mdir, "0.arrow"), "wb") as sink:
        with pa.ipc.new_file(sink, table.schema) as w:
            w.write_table(table, max_chunksize=1 << 20)
    del table
    fast = ArrowMetadataProvider(mdir)
    slow = ReferenceStyleProvider(fast.table)

    res = SimpleNamespace(model=model, tokenizer=tokenizer, preprocess=preprocess, device="cuda:0", image_index=ix, text_index=ix,
                          metadata_is_ordered_by_ivf=False, safety_model=None, violence_detector=None, aesthetic_embeddings=None)
    hp = KnnHotPath()
    rng = np.random.default_rng(0)
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)).save(buf, format="JPEG", quality=90)
    img_b64 = base64.b64encode(buf.getvalue()).decode()
    planted = rng.choice(rows, 64, replace=False)
    qemb = perturbed_queries(ix.reconstruct_batch(np.sort(planted)))

    def timed(fn, reps):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        ts = np.sort(ts)
        return out, float(np.mean(ts)) * 1e3, float(ts[len(ts) // 2]) * 1e3, float(ts[-1]) * 1e3

    stages = {}
    _, m, p50, mx = timed(lambda: hp.compute_query(res, "a photo of the red car on beach at night", None, None, None), a.reps)
    stages["text_query_B1"] = (m, p50, mx)
    _, m, p50, mx = timed(lambda: hp.compute_query(res, None, img_b64, None, None), a.reps)
    stages["image_query_B1 (incl. JPEG decode + preprocess)"] = (m, p50, mx)
    pil = Image.open(io.BytesIO(base64.b64decode(img_b64)))
    px = preprocess(pil).unsqueeze(0)
    _, m, p50, mx = timed(lambda: model.encode_image(px), a.reps)
    stages["image_encode_only_B1"] = (m, p50, mx)
    out = {"rows": rows, "meta_rows": a.meta_rows, "stages_ms": {}, "requests": []}
    for k, dedup in ((40, False), (40, True), (3000, False), (3000, True)):
        j = [0]

        def search():
            j[0] += 1
            return hp.knn_search(qemb[j[0] % 64:j[0] % 64 + 1], "image", k, res, dedup, False, False)

        (dist, ind), m, p50, mx = timed(search, max(3, a.reps // (1 if k <= 64 else 4)))
        stages[f"knn_search k={k}{' +dedup' if dedup else ''}"] = (m, p50, mx)
        ind_m = [np.int64(int(i) % a.meta_rows) for i in ind]  # the metadata table is smaller than the index: fold the ids
        nimg = min(len(ind_m), k)
        _, ms_fast, _, _ = timed(lambda: hp.map_to_metadata(ind_m, dist, nimg, fast, ["url", "caption"]), a.reps)
        _, ms_slow, _, _ = timed(lambda: hp.map_to_metadata(ind_m, dist, nimg, slow, ["url", "caption"]), max(3, a.reps // 4))
        stages[f"map_to_metadata k={k}: batched take"] = (ms_fast, ms_fast, ms_fast)
        stages[f"map_to_metadata k={k}: reference-style per-id slices"] = (ms_slow, ms_slow, ms_slow)
        if not dedup:
            def request():
                q = hp.compute_query(res, "a photo of the red car on beach at night", None, None, None)
                dd, ii = hp.knn_search(q, "image", k, res, True, False, False)
                return hp.map_to_metadata([np.int64(int(i) % a.meta_rows) for i in ii], dd, len(ii), fast, ["url", "caption"])

            _, m, p50, mx = timed(request, max(3, a.reps // (1 if k <= 64 else 4)))
            out["requests"].append({"k": k, "deduplicate": True, "mean_ms": round(m, 3), "p50_ms": round(p50, 3), "max_ms": round(mx, 3)})
            stages[f"WHOLE REQUEST text -> k={k} +dedup -> metadata"] = (m, p50, mx)
    # safety head of the post filter (clip_back.py:315-325; H14 detector stack, h14_nsfw_model.py:16-34) on k result embeddings:
    # GPU (service.Mi355xSafetyHead) vs the reference's torch fp32 module on this box's host cores (random weights: same arithmetic)
    from clip_retrieval_amd.service import Mi355xSafetyHead

    widths, positions = [1024, 1024, 2048, 1024, 256, 128, 16, 1], [0, 3, 6, 9, 12, 15, 16]
    sd, layers = {}, []
    for j, p_ in enumerate(positions):
        lin = torch.nn.Linear(widths[j], widths[j + 1])
        sd[f"layers.{p_}.weight"], sd[f"layers.{p_}.bias"] = lin.weight.data, lin.bias.data
        layers.append(lin)
        if j + 1 < len(positions) and positions[j + 1] != p_ + 1:
            layers += [torch.nn.ReLU(), torch.nn.Dropout(0.2)]
    ref_head = torch.nn.Sequential(*layers).eval()
    head = Mi355xSafetyHead(sd, device=0)
    for k in (40, 3000):
        e = rng.standard_normal((k, 1024)).astype(np.float32)
        _, m, p50, mx = timed(lambda: head.predict(e, batch_size=k), a.reps)
        stages[f"safety head k={k}: GPU (incl. H2D / D2H)"] = (m, p50, mx)

        def ref_predict():
            with torch.no_grad():
                return ref_head(torch.from_numpy(e)).numpy()

        _, m, p50, mx = timed(ref_predict, a.reps)
        stages[f"safety head k={k}: reference torch fp32 on the host"] = (m, p50, mx)
    print(f"{'stage':70s} {'mean ms':>10s} {'p50 ms':>10s} {'max ms':>10s}")
    for name, (m, p50, mx) in stages.items():
        print(f"{name:70s} {m:10.3f} {p50:10.3f} {mx:10.3f}")
        out["stages_ms"][name] = round(m, 3)
    print("REQUEST " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
