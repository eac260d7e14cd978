"""Per-kernel micro-benchmark (for rocprofv3 --pmc passes and A/B of kernel variants): launches the dominant
kernels at their BASELINE shapes a few times each.  Usage: python tools/microbench.py [gemm] [attn] [knn] [ln]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import clip_retrieval_amd  # noqa: E402
from clip_retrieval_amd._lib import check  # noqa: E402

lib = clip_retrieval_amd.load_library()
what = set(sys.argv[1:]) or {"gemm", "attn", "knn", "ln"}  # extra targets: ivf, b1, e2e, reader, pipeline
REPS = int(os.environ.get("MB_REPS", "5"))
P = lambda t: C.c_void_p(t.data_ptr())


def timed(name, fn, flops=None, nbytes=None):
    for _ in range(3):  # (small-batch encodes capture their hipGraph on the second call)
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    st = torch.cuda.current_stream()
    ev[0].record(st)
    for _ in range(REPS):
        fn()
    ev[1].record(st)
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / REPS
    extra = ""
    if flops:
        extra += f"  {flops / ms / 1e9:8.1f} TFLOP/s"
    if nbytes:
        extra += f"  {nbytes / ms / 1e6:8.1f} GB/s"
    print(f"{name:48s} {ms * 1e3:9.1f} us{extra}", flush=True)


# a non-default torch stream: the C ABI treats stream == NULL as "use the handle's own stream", which
# torch.cuda.Event on the default stream would not see
_stream = torch.cuda.Stream()
torch.cuda.set_stream(_stream)
st = _stream.cuda_stream
assert st != 0
def time_once(fn, reps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    s_ = torch.cuda.current_stream()
    ev[0].record(s_)
    for _ in range(reps):
        fn()
    ev[1].record(s_)
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


if "gemm" in what:
    # A/B of GEMM variants: shapes outer, variants interleaved inside each round (same clocks / thermal state),
    # median over MB_ROUNDS rounds.  MB_VARIANTS = "1,3" or "3:dbg" entries (CLIPX_GEMM_DBG ablations of gemm256sp.hip).
    M = 256 * 257
    vspecs = os.environ.get("MB_VARIANTS", "1,3").split(",")
    rounds = int(os.environ.get("MB_ROUNDS", "5"))
    for (name, N, K, epi) in [("qkv  65792x3072x1024", 3072, 1024, 0), ("out  65792x1024x1024", 1024, 1024, 3),
                              ("fc1  65792x4096x1024", 4096, 1024, 1), ("fc2  65792x1024x4096", 1024, 4096, 3),
                              ("txt-fc1 19712x3072x768", 3072, 768, 1), ("peel-qkv 256x3072x1024", 3072, 1024, 0),
                              ("peel-out 256x1024x1024", 1024, 1024, 3), ("peel-fc1 256x4096x1024", 4096, 1024, 1),
                              ("peel-fc2 256x1024x4096", 1024, 4096, 3)]:
        if os.environ.get("MB_GEMM") and not name.startswith(os.environ["MB_GEMM"]):
            continue
        m = 19712 if name.startswith("txt") else (256 if name.startswith("peel") else M)
        A = (torch.randn(m, K, device="cuda") * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        out = torch.zeros(m, N, device="cuda", dtype=torch.float32 if epi == 3 else torch.bfloat16)
        call = lambda: check(lib, lib.clipx_gemm_bf16_device(0, P(A), P(W), P(b), P(out), m, N, K, epi, C.c_void_p(st)), "clipx")

        def setv(vspec):
            os.environ["CLIPX_GEMM_VARIANT"] = vspec.split(":")[0]
            if ":" in vspec:
                os.environ["CLIPX_GEMM_DBG"] = vspec.split(":")[1]
            else:
                os.environ.pop("CLIPX_GEMM_DBG", None)

        setv(vspecs[0])
        time_once(call, 30)  # clock ramp
        res = {v: [] for v in vspecs}
        for _ in range(rounds):
            for v in vspecs:
                setv(v)
                res[v].append(time_once(call, REPS))
        for v in vspecs:
            ms = sorted(res[v])[len(res[v]) // 2]
            print(f"gemm v{v:5s} {name:26s} {ms * 1e3:9.1f} us  {2.0 * m * N * K / ms / 1e9:8.1f} TFLOP/s   (min {min(res[v]) * 1e3:.1f})", flush=True)
if "attn" in what:
    for (B, T, H, causal) in [(256, 257, 16, 0), (256, 77, 12, 1)]:
        qkv = torch.randn(B * T, 3 * H * 64, device="cuda").to(torch.float16)
        out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
        timed(f"attention B={B} T={T} H={H} causal={causal}",
              lambda: check(lib, lib.clipx_attention_device(0, P(qkv), P(out), B, T, H, causal, C.c_void_p(st)), "clipx"),
              flops=4.0 * B * H * T * T * 64)
if "ln" in what:
    M, d = 256 * 257, 1024
    x = torch.randn(M, d, device="cuda")
    g, b = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
    y = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
    timed("layernorm 65792x1024 f32->bf16", lambda: check(lib, lib.clipx_layernorm_device(0, P(x), P(g), P(b), P(y), 1, M, d, C.c_float(1e-5), C.c_void_p(st)), "clipx"),
          nbytes=M * d * 6.0)
if "knn" in what:
    from clip_retrieval_amd.knn import Mi355xIndex

    rows = int(os.environ.get("MB_KNN_ROWS", "20000000"))
    ix = Mi355xIndex(768)
    ix.synth_fill(rows, 3)
    for nq in (1, 32, 64, 128, 256):
        q = torch.nn.functional.normalize(torch.randn(nq, 768, device="cuda"), dim=1)
        D = torch.empty(nq, 40, device="cuda")
        I = torch.empty(nq, 40, device="cuda", dtype=torch.int64)
        s0 = ix.stats()
        ix.profile(True)
        timed(f"knn search_device rows={rows} nq={nq} k=40 (whole call)",
              lambda: ix.search_device(q.data_ptr(), nq, 40, D.data_ptr(), I.data_ptr(), st), nbytes=rows * 768 * 2.0)
        ix.profile(False)
        nl, ms = ix.profile_get()
        s1 = ix.stats()
        print(f"    main scan kernel alone: {ms / max(nl, 1) * 1e3:9.1f} us  {rows * 768 * 2.0 / (ms / max(nl, 1)) / 1e6:8.1f} GB/s   "
              f"proof-served {s1[0] - s0[0]} failed {s1[1] - s0[1]}", flush=True)

if "ivf" in what:
    # IVF-Flat at scale: device-side build (k-means on a resident sample, assignment kernel, streaming scatter) and search.
    # The corpus is a mixture of Gaussians generated on the GPU and copied to host memory chunk by chunk (what a build from
    # img_emb_*.npy files sees); bytes actually scanned by a search = tiles of the probed lists.
    import numpy as np

    from clip_retrieval_amd.knn import Mi355xIndex, build_ivf_index, train_ivf_centroids

    rows, d = int(os.environ.get("MB_IVF_ROWS", "2000000")), int(os.environ.get("MB_IVF_D", "768"))
    nlist = int(os.environ.get("MB_IVF_NLIST", "1024"))
    avail = int(next(l.split()[1] for l in open("/proc/meminfo") if l.startswith("MemAvailable"))) * 1024
    if rows * d * 2 * 1.4 > avail:  # the corpus is held in host memory like a folder of .npy files would be; never drive the box out of memory
        rows = int(avail / (d * 2 * 1.4)) // (1 << 20) * (1 << 20)
        print(f"ivf: host memory allows {rows} rows", flush=True)
    g = torch.Generator(device="cuda").manual_seed(0)
    centers = torch.randn(nlist, d, device="cuda", generator=g)
    x = np.empty((rows, d), dtype=np.float16)
    for o in range(0, rows, 1 << 20):
        m = min(1 << 20, rows - o)
        v = centers[torch.randint(0, nlist, (m,), device="cuda", generator=g)] + 0.5 * torch.randn(m, d, device="cuda", generator=g)
        x[o:o + m] = torch.nn.functional.normalize(v, dim=1).half().cpu().numpy()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cent = train_ivf_centroids(x, nlist, niter=8)
    t1 = time.perf_counter()
    ix = build_ivf_index(x, nlist, nprobe=16, centroids=cent)
    t2 = time.perf_counter()
    print(f"ivf build rows={rows} d={d} nlist={nlist}: k-means (8 iterations on {min(rows, nlist * 256)} rows) {t1 - t0:.1f} s, "
          f"assign + scatter of all rows {t2 - t1:.1f} s, total {t2 - t0:.1f} s", flush=True)
    rng = np.random.default_rng(0)
    nsub = min(rows, int(os.environ.get("MB_IVF_RECALL_ROWS", "1000000")))
    qh = x[rng.integers(0, nsub, 32)].astype(np.float32) + 0.05 * rng.standard_normal((32, d)).astype(np.float32)
    qh /= np.linalg.norm(qh, axis=1, keepdims=True)
    flat = Mi355xIndex(d, coalesce=False)
    flat.add(x)
    _, If = flat.search(qh, 40)
    flat.close()
    for nq, nprobe in ((1, 16), (32, 16), (32, 64), (32, 256)):
        ix.nprobe = min(nprobe, nlist)
        q = torch.from_numpy(qh[:nq]).cuda()
        D = torch.empty(nq, 40, device="cuda")
        I = torch.empty(nq, 40, device="cuda", dtype=torch.int64)
        timed(f"ivf search rows={rows} nlist={nlist} nprobe={nprobe} nq={nq} k=40",
              lambda: ix.search_device(q.data_ptr(), nq, 40, D.data_ptr(), I.data_ptr(), st))
        torch.cuda.synchronize()
        Ii = I.cpu().numpy()
        rec = np.mean([len(set(Ii[i]) & set(If[i])) / 40.0 for i in range(nq)])
        print(f"    recall@40 vs the exact flat scan of the same {rows} rows: {rec:.4f}", flush=True)
if "b1" in what:
    # query-side latency (KnnService.compute_query, clip_back.py:207-255): ONE text / ONE image through the towers
    from clip_retrieval_amd.encoder import ARCHS, ClipEncoder, random_blob
    from clip_retrieval_amd.synth import normalise_u8_nhwc, synth_pixels_u8, synth_tokens

    arch = ARCHS["ViT-L/14"]
    enc = ClipEncoder(arch, random_blob(arch, seed=0), 0)
    pix = torch.from_numpy(normalise_u8_nhwc(synth_pixels_u8(1, arch.image_size, seed=1))).cuda()
    ids = torch.from_numpy(synth_tokens(1, arch.ctx_len, arch.vocab, seed=2)).cuda()
    o16 = torch.empty(1, arch.embed_dim, dtype=torch.float16, device="cuda")
    timed("encode_image B=1 ViT-L/14 (device buffers)", lambda: enc.encode_image_device(pix.data_ptr(), 1, 0, o16.data_ptr(), None, st))
    timed("encode_text  B=1 ViT-L/14 (device buffers)", lambda: enc.encode_text_device(ids.data_ptr(), 1, o16.data_ptr(), None, st))

if "e2e" in what:
    # host-buffer path of the C ABI (what ClipMapper uses): f32 NCHW batches of 256 in host memory -> fp16 embeddings.
    #   device-resident   the kernels alone (inputs already in HBM): the rate bench.py quotes
    #   synchronous call  clipx_encode_image per batch: upload, kernels, download strictly one after the other
    #   tickets           clipx_encode_image_async / clipx_wait, batch n+1 submitted before batch n is collected (what
    #                     runner.Runner does with ClipMapper.submit / collect): the upload hides under the kernels
    import numpy as np

    from clip_retrieval_amd.encoder import ARCHS, ClipEncoder, random_blob
    from clip_retrieval_amd.synth import normalise_u8_nhwc, synth_pixels_u8

    arch = ARCHS[os.environ.get("MB_MODEL", "ViT-L/14")]
    enc = ClipEncoder(arch, random_blob(arch, seed=0), 0)
    NB = int(os.environ.get("MB_E2E_BATCHES", "6"))
    pix = normalise_u8_nhwc(synth_pixels_u8(256, arch.image_size, seed=1))
    pinned = [torch.from_numpy(pix).pin_memory() for _ in range(2)]
    dev = torch.from_numpy(pix).cuda()
    o16 = torch.empty(256, arch.embed_dim, dtype=torch.float16, device="cuda")
    enc.encode_image_device(dev.data_ptr(), 256, 0, o16.data_ptr(), None, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(NB):
        enc.encode_image_device(dev.data_ptr(), 256, 0, o16.data_ptr(), None, st)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / NB
    print(f"encode_image B=256 device-resident: {t_dev * 1e3:.1f} ms/batch = {256 / t_dev:.0f} images/s", flush=True)
    for name, arr in (("pageable numpy", pix), ("page-locked torch tensor", pinned[0])):
        enc.encode_image(arr)
        t0 = time.perf_counter()
        for _ in range(NB):
            enc.encode_image(arr)
        dt = (time.perf_counter() - t0) / NB
        print(f"encode_image B=256 synchronous host call ({name}): {dt * 1e3:.1f} ms/batch = {256 / dt:.0f} images/s "
              f"({t_dev / dt:.3f} of device-resident)", flush=True)
    for name, src in (("page-locked", pinned), ("pageable", [pix, pix])):
        h = enc.submit_image(src[0])
        t0 = time.perf_counter()
        for i in range(1, NB + 1):
            h2 = enc.submit_image(src[i & 1])
            enc.collect(h)
            h = h2
        dt = (time.perf_counter() - t0) / NB
        enc.collect(h)
        print(f"encode_image B=256 tickets, one batch ahead ({name}): {dt * 1e3:.1f} ms/batch = {256 / dt:.0f} images/s "
              f"({t_dev / dt:.3f} of device-resident)", flush=True)

if "reader" in what:
    # reader throughput: a webdataset-style tar of JPEGs -> decode + CLIP transform on a thread pool
    import io
    import tarfile
    import tempfile

    import numpy as np
    from PIL import Image

    from clip_retrieval_amd.reader import HashTokenizer, WebdatasetReader, clip_preprocess, clip_preprocess_u8, decode_rgb_u8
    from clip_retrieval_amd.runner import Sampler

    n = int(os.environ.get("MB_READER_SAMPLES", "4000"))
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "shard.tar")
    with tarfile.open(path, "w") as tf:
        base = rng.integers(0, 255, (256, 256, 3), dtype=np.uint8)
        for i in range(n):
            buf = io.BytesIO()
            Image.fromarray(np.roll(base, i, axis=1)).save(buf, format="JPEG", quality=90)
            for ext, data in (("jpg", buf.getvalue()), ("txt", f"caption number {i}".encode())):
                ti = tarfile.TarInfo(f"{i:06d}.{ext}")
                ti.size = len(data)
                tf.addfile(ti, io.BytesIO(data))
    print(f"host cores: {os.cpu_count()}", flush=True)
    for prep in (clip_preprocess, clip_preprocess_u8, decode_rgb_u8):  # decode_rgb_u8: resize / crop left to the GPU (row f2)
        for procs in (False, True):
            for workers in (8, 32):
                for rep in range(2):  # the second pass reuses the started worker processes (as consecutive partitions do)
                    r = WebdatasetReader(Sampler(0, 1), prep, HashTokenizer(), [path], 256, workers)
                    r.use_processes = procs
                    t0 = time.perf_counter()
                    got = sum(b["image_tensor"].shape[0] if "image_tensor" in b else b["image_raw"]["hw"].shape[0] for b in r)
                    dt = time.perf_counter() - t0
                print(f"WebdatasetReader {got} JPEG 256x256 + captions, {getattr(prep, '__name__', type(prep).__name__)}, {workers} decode "
                      f"{'processes' if procs else 'threads'}: {got / dt:.0f} samples/s", flush=True)
    t0 = time.perf_counter()
    k = sum(1 for _ in WebdatasetReader(Sampler(0, 1), clip_preprocess_u8, HashTokenizer(), [path], 256, 8)._raw_samples())
    print(f"tar iteration alone (parent process): {k / (time.perf_counter() - t0):.0f} samples/s", flush=True)

if "pipeline" in what:
    # The whole drop-in (BASELINE config 4 on one GPU): worker() = tar shards -> WebdatasetReader (decode processes, uint8
    # pixels) -> pipelined Runner -> ClipMapper (async tickets) -> NumpyWriter, ViT-L/14 image + text, random weights.
    import glob
    import gzip
    import io
    import tarfile
    import tempfile

    import numpy as np
    from PIL import Image

    from clip_retrieval_amd.worker import worker

    shards_n, per = int(os.environ.get("MB_PIPE_SHARDS", "8")), int(os.environ.get("MB_PIPE_PER_SHARD", "2048"))
    tmp = tempfile.mkdtemp()
    rng = np.random.default_rng(0)
    jpegs = []
    for i in range(64):
        buf = io.BytesIO()
        Image.fromarray(rng.integers(0, 255, (256, 256, 3), dtype=np.uint8)).save(buf, format="JPEG", quality=90)
        jpegs.append(buf.getvalue())
    shards = []
    for sh in range(shards_n):
        path = os.path.join(tmp, f"{sh:03d}.tar")
        with tarfile.open(path, "w") as tf:
            for i in range(per):
                for ext, data in (("jpg", jpegs[(sh * per + i) % 64]), ("txt", f"a photo number {sh * per + i} of something".encode())):
                    ti = tarfile.TarInfo(f"{sh:03d}{i:06d}.{ext}")
                    ti.size = len(data)
                    tf.addfile(ti, io.BytesIO(data))
        shards.append(path)
    bpe = os.path.join(tmp, "bpe_simple_vocab_16e6.txt.gz")  # a stand-in merges file (the real one is not available offline)
    with gzip.open(bpe, "wt", encoding="utf-8") as f:
        f.write("#version: 0.2\nt h\nth e</w>\no f</w>\np h\n")
    os.environ["CLIP_BPE_PATH"] = bpe
    pipe_model = os.environ.get("MB_PIPE_MODEL", "ViT-L/14")
    if os.environ.get("MB_PIPE_ONE_STREAM"):  # A/B: shards in order through one stream instead of the reference loader's per-worker batching
        from clip_retrieval_amd.reader import WebdatasetReader as _W

        _W.reference_batch_order = False
        print("pipeline: WebdatasetReader.reference_batch_order = False")
    for workers, gpu_resize in ((8, False), (32, False), (8, True), (32, True)):
        out = os.path.join(tmp, f"out{workers}{'r' if gpu_resize else ''}")
        args = dict(input_dataset=shards, output_folder=out, output_partition_count=2, input_format="webdataset", batch_size=256,
                    num_prepro_workers=workers, enable_text=True, enable_image=True, clip_model="random:" + pipe_model,
                    gpu_resize=gpu_resize)
        t0 = time.perf_counter()
        worker([0], **args)  # first partition: model build, worker start-up, warm-up
        t1 = time.perf_counter()
        worker([1], **args)
        t2 = time.perf_counter()
        n = shards_n * per // 2
        rows = sum(np.load(f, mmap_mode="r").shape[0] for f in glob.glob(out + "/img_emb/*.npy"))
        print(f"pipeline worker() {pipe_model} image+text, {workers} decode processes, resize on the {'GPU' if gpu_resize else 'host'}: first partition {n / (t1 - t0):.0f} samples/s "
              f"(with start-up), second partition {n / (t2 - t1):.0f} samples/s; {rows} embeddings written", flush=True)
        for f in sorted(glob.glob(out + "/stats/*.json"))[-1:]:
            print("    stats of the last partition:", open(f).read()[:400], flush=True)

if "sharded" in what:
    # The one-process sharded index (knnx_shards_*, what KnnService holds on a multi-GPU node) against the single index over
    # the same rows -- here with all shards on ONE GPU, so the scans serialise: what is measured is the price of the shard
    # fan-out (query copies, per-shard top-k, peer gather, merge), not a scaling curve.
    import numpy as np

    from clip_retrieval_amd.knn import Mi355xIndex, ShardedMi355xIndex

    total = int(os.environ.get("MB_SHARD_ROWS", "100000000"))
    d, k = 768, 40
    rng = np.random.default_rng(0)
    q = rng.standard_normal((64, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)

    def host_timed(name, fn, reps=5):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name:70s} {dt * 1e3:8.2f} ms per call  {64 / dt:8.0f} QPS", flush=True)

    one = Mi355xIndex(d, coalesce=False)
    one.synth_fill(total, 3)
    host_timed(f"Mi355xIndex.search 64 queries, {total} rows, host buffers", lambda: one.search(q, k))
    one.close()
    for ns in (2, 4, 8):
        sh = ShardedMi355xIndex(d, [0] * ns, coalesce=False)
        sh.synth_fill(total // ns, 3)
        host_timed(f"ShardedMi355xIndex.search 64 queries, {ns} shards x {total // ns} rows on one GPU", lambda: sh.search(q, k))
        sh.close()
