"""GPU tests of the product code round 2 left unexercised (VERDICT r2, next-round item 3 a / b / e):
checkpoint + vocabulary loading through `ClipMapper(clip_model=..., clip_cache_path=...)` / `load_clip`, the request path
`KnnHotPath.compute_query -> knn_search -> map_to_metadata`, and `worker.worker()` itself over tar shards."""
import base64
import gzip
import io
import json
import os
import tarfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
COS_BAR = 1.0 - 1e-3


def _cos(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def _openai_state_dict(W, arch):
    """The oracle's unpacked weights under OpenAI `clip` / open_clip state-dict names (visual.proj and text_projection are
    stored [width, embed] there)."""
    sd = {"visual.conv1.weight": W["conv"].reshape(arch.v_width, 3, arch.patch_size, arch.patch_size), "visual.class_embedding": W["cls"],
          "visual.positional_embedding": W["vpos"], "visual.ln_pre.weight": W["ln_pre_w"], "visual.ln_pre.bias": W["ln_pre_b"],
          "visual.ln_post.weight": W["ln_post_w"], "visual.ln_post.bias": W["ln_post_b"], "visual.proj": W["vproj"].T.contiguous(),
          "token_embedding.weight": W["tok"], "positional_embedding": W["tpos"], "ln_final.weight": W["ln_final_w"],
          "ln_final.bias": W["ln_final_b"], "text_projection": W["tproj"].T.contiguous(), "logit_scale": torch.tensor(4.6)}
    names = {"ln1_w": "ln_1.weight", "ln1_b": "ln_1.bias", "qkv_w": "attn.in_proj_weight", "qkv_b": "attn.in_proj_bias",
             "out_w": "attn.out_proj.weight", "out_b": "attn.out_proj.bias", "ln2_w": "ln_2.weight", "ln2_b": "ln_2.bias",
             "fc1_w": "mlp.c_fc.weight", "fc1_b": "mlp.c_fc.bias", "fc2_w": "mlp.c_proj.weight", "fc2_b": "mlp.c_proj.bias"}
    for prefix, layers in (("visual.transformer", W["vlayers"]), ("transformer", W["tlayers"])):
        for i, L in enumerate(layers):
            for k, n in names.items():
                sd[f"{prefix}.resblocks.{i}.{n}"] = L[k]
    return {k: v.clone() for k, v in sd.items()}


def _merges_file(folder):
    """A CLIP-format merges file (synthetic: the real one is not available offline) under the name find_bpe_file looks for."""
    from clip_retrieval_amd.tokenizer import BPE_FILE_NAME, bytes_to_unicode

    b2u = bytes_to_unicode()
    pairs = []
    for w in ("photo", "cat", "dog", "the", "of", "a"):
        sym = [b2u[b] for b in w.encode()]
        sym[-1] += "</w>"
        while len(sym) > 1:
            pairs.append((sym[0], sym[1]))
            sym = [sym[0] + sym[1]] + sym[2:]
    seen, merges = set(), []
    for p in pairs:
        if p not in seen:
            seen.add(p)
            merges.append(p)
    path = os.path.join(folder, BPE_FILE_NAME)
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    return path


@pytest.fixture(scope="module")
def tiny_checkpoints(tmp_path_factory):
    """tmp/<hf|openai>/tiny-test.{safetensors,pt} of the tiny ViT-B/32-shaped oracle + a merges file beside each."""
    from safetensors.torch import save_file

    import clip_retrieval_amd.encoder as E
    from oracle.clip_oracle import ARCHS, HFClipOracle, unpack_blob

    arch = ARCHS["tiny-B/32"]
    oracle = HFClipOracle(arch, seed=0)
    root = tmp_path_factory.mktemp("ckpt")
    hf, oa = root / "hf", root / "openai"
    hf.mkdir()
    oa.mkdir()
    sd = {k: v.contiguous() for k, v in oracle.model.state_dict().items() if v.dtype.is_floating_point}
    save_file(sd, str(hf / "tiny-test.safetensors"))
    W = unpack_blob(torch.from_numpy(oracle.export_blob()), arch)
    torch.save({"state_dict": {"module." + k: v for k, v in _openai_state_dict(W, arch).items()}}, str(oa / "tiny-test.pt"))
    for d in (hf, oa):
        _merges_file(str(d))
    E.ARCHS["tiny-test"] = E.ClipArch(**{k: getattr(arch, k) for k in E.ClipArch.__dataclass_fields__})
    yield arch, oracle, {"hf": str(hf), "openai": str(oa)}
    E.ARCHS.pop("tiny-test", None)
    with E._registry_lock:  # pylint: disable=protected-access
        for key in [k for k in E._registry if k[0] == "tiny-test"]:  # pylint: disable=protected-access
            E._registry.pop(key).close()  # pylint: disable=protected-access


@pytest.mark.parametrize("kind", ["hf", "openai"])
def test_checkpoint_and_vocabulary_are_loaded_from_clip_cache_path(tiny_checkpoints, kind):
    """mapper.py:36-41 `load_clip(clip_model, clip_cache_path=...)`: the weights come from a file on disk (HF-keyed
    .safetensors / OpenAI-keyed .pt saved from a wrapper with a `module.` prefix) and the tokenizer from the merges file next to
    it; `ClipMapper.__call__` on real caption strings and image tensors must match the oracle that owns those weights."""
    import clip_retrieval_amd.encoder as E
    from clip_retrieval_amd.mapper import ClipMapper
    from oracle.clip_oracle import mapper_semantics, normalise_u8_nhwc, synth_pixels_u8

    arch, oracle, dirs = tiny_checkpoints
    with E._registry_lock:  # pylint: disable=protected-access
        for key in [k for k in E._registry if k[0] == "tiny-test"]:  # the two checkpoints share the model name
            E._registry.pop(key).close()  # pylint: disable=protected-access
    model, preprocess, tokenizer = E.load_clip("tiny-test", use_jit=False, warmup_batch_size=1, clip_cache_path=dirs[kind])
    caps = ["a photo of a cat", "the dog of the cat", "a"]
    tokens = tokenizer(caps)
    assert tuple(tokens.shape) == (3, arch.ctx_len) and int(tokens[0, 0]) == tokenizer.sot_token and int(tokens.max()) == tokenizer.eot_token
    pix = torch.from_numpy(normalise_u8_nhwc(synth_pixels_u8(3, seed=4)))
    mapper = ClipMapper(enable_image=True, enable_text=True, enable_metadata=False, use_mclip=False, clip_model="tiny-test",
                        use_jit=False, mclip_model="", warmup_batch_size=1, clip_cache_path=dirs[kind])
    out = mapper({"image_tensor": pix, "text_tokens": tokens, "image_filename": ["a", "b", "c"], "text": caps, "metadata": None})
    _, wi = mapper_semantics(oracle.encode_image(pix))
    _, wt = mapper_semantics(oracle.encode_text(tokens))
    assert out["image_embs"].dtype == np.float16 and out["image_embs"].shape == (3, arch.embed_dim)
    assert _cos(out["image_embs"], wi).min() >= COS_BAR and _cos(out["text_embs"], wt).min() >= COS_BAR, kind
    assert out["text"] == caps
    ft = model.encode_text(tokens)
    assert ft.dtype == torch.float32 and _cos(ft.numpy(), wt).min() >= COS_BAR


def test_request_path_compute_query_knn_search_map_to_metadata(tiny_checkpoints, tmp_path):
    """One /knn-service request end to end on the GPU (clip_back.py:419-470 = compute_query -> knn_search -> map_to_metadata):
    text, base64-image and embedding queries with the aesthetic shift; the index holds the oracle's embeddings of 300 synthetic
    images + captions, so the expected neighbours come from the numpy oracle; metadata through ArrowMetadataProvider."""
    from types import SimpleNamespace

    import pyarrow as pa
    from PIL import Image

    import clip_retrieval_amd.encoder as E
    from clip_retrieval_amd.knn import Mi355xIndex
    from clip_retrieval_amd.reader import clip_preprocess
    from clip_retrieval_amd.service import ArrowMetadataProvider, KnnHotPath
    from oracle.clip_oracle import mapper_semantics, synth_pixels_u8
    from oracle.knn_oracle import FlatIPOracle

    arch, oracle, dirs = tiny_checkpoints
    model, preprocess, tokenizer = E.load_clip("tiny-test", use_jit=False, warmup_batch_size=1, clip_cache_path=dirs["hf"])
    n = 300
    u8 = synth_pixels_u8(n, arch.image_size, seed=11)
    pix = torch.from_numpy(np.stack([clip_preprocess(Image.fromarray(im), size=arch.image_size) for im in u8]))
    emb16, _ = mapper_semantics(oracle.encode_image(pix))
    ix, ora = Mi355xIndex(arch.embed_dim), FlatIPOracle(arch.embed_dim)
    ix.add(emb16)
    ora.add(emb16)
    t = pa.table({"url": [f"http://x/{i}" for i in range(n)], "caption": [f"caption {i}".encode() for i in range(n)]})
    with pa.OSFile(str(tmp_path / "0.arrow"), "wb") as sink:
        with pa.ipc.new_file(sink, t.schema) as w:
            w.write_table(t)
    provider = ArrowMetadataProvider(str(tmp_path))
    aest = np.random.default_rng(0).standard_normal((10, arch.embed_dim)).astype(np.float32)
    aest /= np.linalg.norm(aest, axis=1, keepdims=True)
    prompts = np.stack([emb16[0].astype(np.float32), emb16[5].astype(np.float32)])  # "violent" = looks like image 5
    res = SimpleNamespace(model=model, tokenizer=tokenizer, preprocess=preprocess, device="cuda:0", image_index=ix, text_index=ix,
                          metadata_is_ordered_by_ivf=False, safety_model=None, violence_detector=prompts, aesthetic_embeddings=aest)
    hp = KnnHotPath()

    # image query: the PNG of image 7, base64 like the front-end sends it
    buf = io.BytesIO()
    Image.fromarray(u8[7]).save(buf, format="PNG")
    q = hp.compute_query(res, None, base64.b64encode(buf.getvalue()).decode(), None, None, False, None, None)
    _, want = mapper_semantics(oracle.encode_image(pix[7:8]))
    assert q.shape == (1, arch.embed_dim) and q.dtype == np.float32 and _cos(q, want).min() >= COS_BAR
    dist, ind = hp.knn_search(q, "image", 10, res, False, False, False)
    Do, Io = ora.search(q, 10)
    assert [int(i) for i in ind] == Io[0].tolist() and int(ind[0]) == 7 and np.allclose(dist, Do[0], atol=1e-5)
    recs = hp.map_to_metadata(ind, dist, 4, provider, ["url", "caption"])
    assert len(recs) == 10 and recs[0] == {"url": "http://x/7", "caption": "caption 7", "id": 7, "similarity": float(dist[0])}
    assert set(recs[3]) == {"url", "caption", "id", "similarity"} and set(recs[4]) == {"id", "similarity"}

    # text query with the aesthetic shift (clip_back.py:251-254)
    q = hp.compute_query(res, "a photo of a cat", None, None, None, False, 9, 0.5)
    _, wt = mapper_semantics(oracle.encode_text(tokenizer(["a photo of a cat"])))
    wq = wt + aest[9] * 0.5
    wq /= np.linalg.norm(wq)
    assert _cos(q, wq).min() >= COS_BAR and abs(np.linalg.norm(q) - 1) < 1e-5
    dist, ind = hp.knn_search(q, "text", 8, res, False, False, False)
    Do, Io = ora.search(q, 8)
    assert [int(i) for i in ind] == Io[0].tolist()

    # embedding query, violence filter on: results that look more like prompt 1 (image 5) than prompt 0 (image 0) are dropped
    q = hp.compute_query(res, None, None, None, emb16[5].astype(np.float32).tolist(), False, None, None)
    dist, ind = hp.knn_search(q, "image", 40, res, False, False, True)
    Do, Io, Ro = ora.search_and_reconstruct(q, 40)
    Rn = Ro[0] / np.linalg.norm(Ro[0], axis=1, keepdims=True)
    s = Rn @ prompts.T
    keep = [int(i) for i, row in zip(Io[0], s) if not (np.argmax(row) == 1 and abs(row[0] - row[1]) > 1e-3)]
    maybe = [int(i) for i, row in zip(Io[0], s) if abs(row[0] - row[1]) <= 1e-3]
    got = [int(i) for i in ind]
    assert 5 not in got and [i for i in got if i not in maybe] == [i for i in keep if i not in maybe]
    ix.close()


def test_worker_on_the_gpu_over_tar_shards(tiny_checkpoints, tmp_path):
    """worker.worker() as the reference runs it (worker.py:22-127): tar shards of JPEGs + captions -> WebdatasetReader (uint8
    pixels, normalised on the GPU) -> pipelined Runner -> ClipMapper -> NumpyWriter.  Every written embedding must match the
    oracle on the SAME decoded pixels and the same tokens."""
    import pandas as pd
    from PIL import Image

    from clip_retrieval_amd.reader import clip_preprocess
    from clip_retrieval_amd.worker import worker
    from oracle.clip_oracle import mapper_semantics

    arch, oracle, dirs = tiny_checkpoints
    rng = np.random.default_rng(0)
    shards, jpegs, k = [], [], 0
    for t in range(2):
        p = tmp_path / f"{t:03d}.tar"
        with tarfile.open(p, "w") as tf:
            for _ in range(9):
                hh, ww = ((96, 96), (70, 131), (150, 64), (33, 47))[k % 4]  # sources of different sizes and aspect ratios
                gy, gx = np.linspace(0, 255, hh, dtype=np.float32), np.linspace(0, 255, ww, dtype=np.float32)
                img = (gx[None, :, None] * 0.5 + gy[:, None, None] * 0.5 + rng.normal(0, 8, (hh, ww, 3))).clip(0, 255).astype(np.uint8)
                buf = io.BytesIO()
                if k % 9 == 7:    # a palette image (Pillow resamples mode P with NEAREST) and
                    Image.fromarray(img).quantize(32).save(buf, format="PNG")
                elif k % 9 == 8:  # an image with an alpha channel (premultiplied resampling): ADVICE r3 -- the reference resizes in the
                    a = np.linspace(0, 255, hh * ww, dtype=np.float32).reshape(hh, ww).astype(np.uint8)  # image's own mode, THEN converts
                    Image.merge("RGBA", (*Image.fromarray(img).split(), Image.fromarray(a))).save(buf, format="PNG")
                else:
                    Image.fromarray(img).save(buf, format="JPEG", quality=92)
                jpegs.append(buf.getvalue())
                for ext, data in (("jpg", buf.getvalue()), ("txt", ("a photo of a cat" if k % 2 else "the dog").encode())):
                    ti = tarfile.TarInfo(f"{k:06d}.{ext}")
                    ti.size = len(data)
                    tf.addfile(ti, io.BytesIO(data))
                k += 1
        shards.append(str(p))
    out = tmp_path / "out"
    worker(tasks=[0, 1], input_dataset=shards, output_folder=str(out), output_partition_count=2, input_format="webdataset",
           batch_size=4, num_prepro_workers=2, enable_text=True, enable_image=True, enable_metadata=False, clip_model="tiny-test",
           clip_cache_path=dirs["openai"], device=0, gpu_normalise=True)
    from clip_retrieval_amd.tokenizer import SimpleTokenizer

    tok = SimpleTokenizer(clip_cache_path=dirs["openai"], context_length=arch.ctx_len)
    for i in range(2):
        img = np.load(out / "img_emb" / f"img_emb_{i}.npy")
        txt = np.load(out / "text_emb" / f"text_emb_{i}.npy")
        meta = pd.read_parquet(out / "metadata" / f"metadata_{i}.parquet")
        assert img.shape == (9, arch.embed_dim) == txt.shape and len(meta) == 9
        pix = torch.from_numpy(np.stack([clip_preprocess(Image.open(io.BytesIO(j)), size=arch.image_size) for j in jpegs[9 * i:9 * i + 9]]))
        _, wi = mapper_semantics(oracle.encode_image(pix))
        _, wt = mapper_semantics(oracle.encode_text(tok(list(meta["caption"]))))
        assert _cos(img, wi).min() >= COS_BAR and _cos(txt, wt).min() >= COS_BAR
        assert json.loads((out / "stats" / f"{i}.json").read_text())["sample_count"] == 9
    # the same job with the resize + centre crop on the GPU as well (row f2): the decode processes hand over the decoded sources
    # (the palette and RGBA members: the host transform's ready crop, reader.DecodeRgbU8); the kernel is bit-identical to Pillow, so
    # every written embedding must be the SAME BYTES as above
    out2 = tmp_path / "out_gpu_resize"
    worker(tasks=[0, 1], input_dataset=shards, output_folder=str(out2), output_partition_count=2, input_format="webdataset",
           batch_size=4, num_prepro_workers=2, enable_text=True, enable_image=True, enable_metadata=False, clip_model="tiny-test",
           clip_cache_path=dirs["openai"], device=0, gpu_resize=True)
    for i in range(2):
        assert np.array_equal(np.load(out / "img_emb" / f"img_emb_{i}.npy"), np.load(out2 / "img_emb" / f"img_emb_{i}.npy"))
        assert np.array_equal(np.load(out / "text_emb" / f"text_emb_{i}.npy"), np.load(out2 / "text_emb" / f"text_emb_{i}.npy"))
        assert json.loads((out2 / "stats" / f"{i}.json").read_text())["sample_count"] == 9


def test_gpu_worker_under_the_launcher_environment(tiny_checkpoints, tmp_path, monkeypatch):
    """worker.gpu_worker() as one rank of a `torch.distributed.run` launch: RANK / LOCAL_RANK / WORLD_SIZE from the
    environment, the output partitions dealt with get_task_list (slurm_worker.py:16-37), the process bound to GPU LOCAL_RANK, then
    worker.worker() -- no collective anywhere.  World 1: both partitions, the same bytes as a direct worker() call; rank 1 of a
    world of 2 (still GPU 0 here): partition 1 only."""
    from PIL import Image

    from clip_retrieval_amd.worker import gpu_worker, worker

    _, _, dirs = tiny_checkpoints
    rng = np.random.default_rng(5)
    shards, k = [], 0
    for t in range(2):
        p = tmp_path / f"{t:03d}.tar"
        with tarfile.open(p, "w") as tf:
            for _ in range(5):
                img = rng.integers(0, 256, (48 + 8 * (k % 3), 64, 3), dtype=np.uint8)
                buf = io.BytesIO()
                Image.fromarray(img).save(buf, format="JPEG", quality=90)
                for ext, data in (("jpg", buf.getvalue()), ("txt", ("a photo of a cat" if k % 2 else "the dog").encode())):
                    ti = tarfile.TarInfo(f"{k:06d}.{ext}")
                    ti.size = len(data)
                    tf.addfile(ti, io.BytesIO(data))
                k += 1
        shards.append(str(p))
    common = dict(input_dataset=shards, output_partition_count=2, input_format="webdataset", batch_size=4, num_prepro_workers=2,
                  enable_text=True, enable_image=True, enable_metadata=False, clip_model="tiny-test", clip_cache_path=dirs["openai"])
    worker(tasks=[0, 1], output_folder=str(tmp_path / "direct"), device=0, **common)
    for key in ("SLURM_PROCID", "SLURM_LOCALID", "WORKER_ARGS_PATH", "NUM_TASKS"):
        monkeypatch.delenv(key, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    gpu_worker(output_folder=str(tmp_path / "rank0of1"), **common)
    for i in range(2):
        for sub, name in (("img_emb", f"img_emb_{i}.npy"), ("text_emb", f"text_emb_{i}.npy")):
            a, b = np.load(tmp_path / "direct" / sub / name), np.load(tmp_path / "rank0of1" / sub / name)
            assert a.shape[0] == 5
            assert np.array_equal(a, b), (sub, name)
        assert json.loads((tmp_path / "rank0of1" / "stats" / f"{i}.json").read_text())["sample_count"] == 5
    assert torch.cuda.current_device() == 0
    # rank 1 of 2 (both "GPUs" are device 0 on this box: LOCAL_RANK stays 0): only its own partition is written
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    gpu_worker(output_folder=str(tmp_path / "rank1of2"), **common)
    assert sorted(os.listdir(tmp_path / "rank1of2" / "img_emb")) == ["img_emb_1.npy"]
    assert np.array_equal(np.load(tmp_path / "rank1of2" / "img_emb" / "img_emb_1.npy"), np.load(tmp_path / "direct" / "img_emb" / "img_emb_1.npy"))


def _h14_state_dict(seed=0, input_size=1024):
    """A state dict with the H14 detector's keys and shapes (h14_nsfw_model.py:16-34), seeded random weights scaled like
    torch's Linear init so the activations keep O(1) magnitude through the stack."""
    rng = np.random.default_rng(seed)
    widths = [input_size, 1024, 2048, 1024, 256, 128, 16, 1]
    positions = [0, 3, 6, 9, 12, 15, 16]
    sd = {}
    for p, (i, o) in zip(positions, zip(widths[:-1], widths[1:])):
        sd[f"layers.{p}.weight"] = (rng.uniform(-1, 1, (o, i)) * np.sqrt(3.0 / i)).astype(np.float32)
        sd[f"layers.{p}.bias"] = (rng.uniform(-1, 1, o) * 0.1).astype(np.float32)
    return sd, positions


def _h14_forward_f64(sd, positions, x):
    """h14_nsfw_model.py:16-34 + 40-42 restated in float64: Linear, ReLU (Dropout = identity in eval) ... Linear(128, 16),
    Linear(16, 1) with no activation between the last two."""
    y = x.astype(np.float64)
    for j, p in enumerate(positions):
        y = y @ sd[f"layers.{p}.weight"].astype(np.float64).T + sd[f"layers.{p}.bias"].astype(np.float64)
        if j + 1 < len(positions) and positions[j + 1] != p + 1:
            y = np.maximum(y, 0)
    return y


@pytest.mark.parametrize("n", [1, 37, 3000])
def test_safety_head_on_the_gpu_matches_the_h14_detector(n):
    """`Mi355xSafetyHead.predict` == the reference detector's forward (torch fp32 on the CPU, and a float64 restatement) on the
    same state dict; `KnnHotPath.get_unsafe_items` flags the same rows."""
    from clip_retrieval_amd.service import KnnHotPath, Mi355xSafetyHead

    sd, positions = _h14_state_dict(seed=n)
    rng = np.random.default_rng(n + 1)
    x = rng.standard_normal((n, 1024)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x *= 30.0
    # centre the outputs on the 0.5 threshold so that both classes occur
    sd["layers.16.bias"] = (sd["layers.16.bias"] + 0.5 - np.median(_h14_forward_f64(sd, positions, x))).astype(np.float32)
    head = Mi355xSafetyHead({k: torch.from_numpy(v) for k, v in sd.items()}, device=0)
    assert head.dims == [1024, 1024, 2048, 1024, 256, 128, 16, 1] and head.relu == [True] * 5 + [False, False]
    got = head.predict(x, batch_size=n)
    want = _h14_forward_f64(sd, positions, x)
    assert got.shape == (n, 1) and got.dtype == np.float32
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() <= 2e-5 * scale, (np.abs(got - want).max(), scale)
    # the reference module itself (same layer stack, eval mode), fp32 on the CPU
    layers = []
    widths = head.dims
    for j, p in enumerate(positions):
        lin = torch.nn.Linear(widths[j], widths[j + 1])
        lin.weight.data = torch.from_numpy(sd[f"layers.{p}.weight"])
        lin.bias.data = torch.from_numpy(sd[f"layers.{p}.bias"])
        layers.append(lin)
        if head.relu[j]:
            layers += [torch.nn.ReLU(), torch.nn.Dropout(0.2)]
    ref = torch.nn.Sequential(*layers).eval()
    with torch.no_grad():
        ref_y = ref(torch.from_numpy(x)).numpy()
    assert np.abs(got - ref_y).max() <= 4e-5 * scale

    class RefModel:
        def predict(self, e, batch_size):
            with torch.no_grad():
                return ref(torch.from_numpy(e)).numpy()

    hp = KnnHotPath()
    a, b = hp.get_unsafe_items(head, x), hp.get_unsafe_items(RefModel(), x)
    near = np.flatnonzero(np.abs(want[:, 0] - 0.5) < 1e-3)  # rows on the threshold may legitimately land on either side
    assert set(a) - set(near) == set(b) - set(near)
    if n >= 37:
        assert 0 < len(b) < n  # both classes occur: the comparison means something
    head.close()


def test_safety_head_arguments():
    from clip_retrieval_amd._lib import HipLibraryError
    from clip_retrieval_amd.service import Mi355xSafetyHead

    sd, _ = _h14_state_dict(seed=1, input_size=64)
    head = Mi355xSafetyHead(sd, device=0)
    assert head.predict(np.zeros((0, 64), dtype=np.float32)).shape == (0, 1)
    with pytest.raises(ValueError):
        head.predict(np.zeros((3, 65), dtype=np.float32))
    head.close()
    with pytest.raises(ValueError):
        Mi355xSafetyHead({"layers.0.weight": np.zeros((8, 4), np.float32), "layers.1.weight": np.zeros((2, 9), np.float32)})
    with pytest.raises(ValueError):
        Mi355xSafetyHead({})
    with pytest.raises(FileNotFoundError):
        Mi355xSafetyHead.from_cache("/nonexistent-cache")


def test_violence_filter_is_the_fp32_einsum_and_follows_a_swapped_detector():
    """clip_back.py:327-331 on the GPU in fp32 (the round-3 form stored the prompts as fp16 rows, cached them under
    id(array) without holding the array, and held a lock across the GPU call): same picks as the reference's numpy einsum on
    random vectors INCLUDING near-ties that fp16 prompts would flip, a replaced detector is picked up even when it is a new
    array of the same shape, and concurrent callers agree."""
    from concurrent.futures import ThreadPoolExecutor

    from clip_retrieval_amd.service import KnnHotPath

    rng = np.random.default_rng(0)
    d = 768
    hp = KnnHotPath()
    for trial in range(3):
        prompts = rng.standard_normal((2, d)).astype(np.float32)
        prompts /= np.linalg.norm(prompts, axis=1, keepdims=True)
        emb = rng.standard_normal((400, d)).astype(np.float32)
        emb /= np.linalg.norm(emb, axis=1, keepdims=True)
        # plant near-ties: rows whose two scores differ by ~1e-5 (an fp16 prompt matrix has errors of ~1e-4 per score)
        diff = prompts[1] - prompts[0]
        for i in range(0, 100):
            e = emb[i] - (emb[i] @ diff) / (diff @ diff) * diff          # exactly tied
            emb[i] = e + (1e-5 if i % 2 else -1e-5) * diff / (diff @ diff)
        want_scores = np.einsum("ij,kj->ik", emb, prompts)
        want = np.where(np.argmax(want_scores, axis=1) == 1)[0]
        margin = np.abs(want_scores[:, 1] - want_scores[:, 0])
        got = hp.get_violent_items(prompts, emb)
        clear = margin > 2e-6  # beyond f32 summation-order noise the pick is the reference's
        assert np.array_equal(np.intersect1d(got, np.flatnonzero(clear)), np.intersect1d(want, np.flatnonzero(clear)))
        assert (margin[:100] < 5e-5).all() and clear[:100].sum() > 80  # the planted near-ties were really tested
        with ThreadPoolExecutor(8) as ex:
            for g in ex.map(lambda _: hp.get_violent_items(prompts, emb), range(16)):
                assert np.array_equal(g, got)
        swapped = prompts[::-1].copy()  # "violent" and "safe" exchanged: the complement on the clear rows
        got2 = hp.get_violent_items(swapped, emb)
        assert np.array_equal(np.intersect1d(got2, np.flatnonzero(clear)), np.setdiff1d(np.flatnonzero(clear), want))
    assert len(hp._prompts) <= 8  # pylint: disable=protected-access
    assert hp.get_violent_items(prompts, np.zeros((0, d), np.float32)).shape == (0,)


def test_knn_search_for_100000_results_widens_the_ivf_probe_like_the_reference():
    """clip_back.py:356-369: num_result_ids >= 100000 sets nprobe = ceil(k / 3000) for the search and restores it.  400 k rows in
    64 lists at nprobe 2 reach ~12 k rows; the request must come back with 100 000 distinct ids, best first, and nprobe as before."""
    from types import SimpleNamespace

    from clip_retrieval_amd.knn import build_ivf_index
    from clip_retrieval_amd.service import KnnHotPath

    rng = np.random.default_rng(3)
    d, n, nlist = 256, 400_000, 64
    x = rng.standard_normal((n, d)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
    ix = build_ivf_index(x, nlist, nprobe=2, niter=2, seed=0)
    res = SimpleNamespace(image_index=ix, text_index=ix, metadata_is_ordered_by_ivf=False, safety_model=None, violence_detector=None)
    hp = KnnHotPath()
    q = x[7:8].astype(np.float32)
    dist, ids = hp.knn_search(q, "image", 3000, res, deduplicate=False, use_safety_model=False, use_violence_detector=False)
    assert len(ids) == 3000 and ix.nprobe == 2
    dist, ids = hp.knn_search(q, "image", 100_000, res, deduplicate=False, use_safety_model=False, use_violence_detector=False)
    assert ix.nprobe == 2, "nprobe must be restored after the request"
    assert len(ids) == 100_000 == len(set(int(i) for i in ids)) and ids[0] == 7 and all(a >= b for a, b in zip(dist, dist[1:]))
    # the answer is the exact top-100 000 of the rows in the 34 probed lists
    ix.nprobe = 34
    D, I = ix.search(q, 100_000)
    ix.nprobe = 2
    assert np.array_equal(np.asarray(ids), I[0])
    ix.close()
