"""Host-side mirrors of the reference interfaces vs golden vectors produced by the reference's own code
(tests/golden/make_golden.py) and vs the expectations of the reference's tests."""
import functools
import hashlib
import io
import json
import os
import tarfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_host_logic.json")))


def test_sampler_matches_reference_runner():
    from clip_retrieval_amd.runner import Sampler

    for g in GOLD["sampler"]:
        items = [f"k{i:03d}" for i in range(g["n"])]
        assert Sampler(g["id"], g["count"])(items) == g["out"]


def test_get_task_list_matches_reference_tests():
    from clip_retrieval_amd.runner import get_task_list

    for g in GOLD["get_task_list"]:
        for r, want in enumerate(g["out"]):
            assert get_task_list(g["num_tasks"], g["world_size"], r, -1) == want
    # every task exactly once for any split
    for n in range(0, 40):
        for w in range(1, 9):
            got = sum((get_task_list(n, w, r) for r in range(w)), [])
            assert got == list(range(n))


def test_writer_bytes_match_reference_writer(tmp_path):
    import pandas as pd
    from clip_retrieval_amd.writer import NumpyWriter

    for g in GOLD["writer"]:
        out = tmp_path / f"p{g['partition_id']}"
        w = NumpyWriter(g["partition_id"], str(out), True, True, True, g["partition_count"])
        for b in g["batches"]:
            b = dict(b)
            b["image_embs"] = np.asarray(b["image_embs"], dtype=np.float16)
            b["text_embs"] = np.asarray(b["text_embs"], dtype=np.float16)
            w(b)
        w.flush()
        seen = {}
        for root, _, names in os.walk(out):
            for name in names:
                seen[os.path.relpath(os.path.join(root, name), out)] = os.path.join(root, name)
        assert sorted(seen) == sorted(g["files"])
        for rel, want in g["files"].items():
            if rel.endswith(".npy"):
                assert hashlib.sha256(open(seen[rel], "rb").read()).hexdigest() == want, rel
            else:
                df = pd.read_parquet(seen[rel])
                assert list(df.columns) == want["columns"]
                assert json.loads(df.to_json(orient="records")) == want["rows"]


def test_writer_empty_partition_writes_nothing(tmp_path):
    from clip_retrieval_amd.writer import NumpyWriter

    w = NumpyWriter(0, str(tmp_path), True, True, False, 4)
    w.flush()
    assert not list((tmp_path / "img_emb").glob("*"))


def test_normalized_zero_guard():
    from clip_retrieval_amd.mapper import normalized

    a = np.array([[3.0, 4.0], [0.0, 0.0]], dtype=np.float32)
    n = normalized(a)
    assert np.allclose(n[0], [0.6, 0.8]) and np.array_equal(n[1], [0, 0])


def _jpeg(w, h, seed):
    from PIL import Image

    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), rng.integers(0, 255, (h, w))], -1).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG")
    return buf.getvalue()


@pytest.fixture()
def image_folder(tmp_path):
    # same population as the reference fixture folder: 7 JPEGs of assorted sizes (test_reader.py:58)
    sizes = [(123, 456), (208, 495), (321, 421), (389, 535), (416, 264), (456, 123), (524, 316)]
    d = tmp_path / "imgs"
    d.mkdir()
    for i, (w, h) in enumerate(sizes):
        (d / f"{w}_{h}.jpg").write_bytes(_jpeg(w, h, i))
    return str(d)


@pytest.fixture()
def tar_shards(tmp_path):
    # same member counts as the reference tars: 4 + 3 + 2 + 2 images (test_reader.py:60)
    paths, k = [], 0
    for t, n in enumerate([4, 3, 2, 2]):
        p = tmp_path / f"image{t + 1}.tar"
        with tarfile.open(p, "w") as tf:
            for _ in range(n):
                data = _jpeg(200 + 10 * k, 150 + 7 * k, k)
                ti = tarfile.TarInfo(f"{k:05d}.jpg")
                ti.size = len(data)
                tf.addfile(ti, io.BytesIO(data))
                cap = f"caption number {k}".encode()
                ti = tarfile.TarInfo(f"{k:05d}.txt")
                ti.size = len(cap)
                tf.addfile(ti, io.BytesIO(cap))
                k += 1
        paths.append(str(p))
    return paths


def test_files_reader_batch_shapes(image_folder):
    """Reference expectation: 7 images, 2 partitions, batch 2 -> [[2,2],[2,1]] (test_reader.py:58-59)."""
    from clip_retrieval_amd.reader import FilesReader
    from clip_retrieval_amd.runner import Sampler

    got = []
    for pid in range(2):
        r = FilesReader(Sampler(pid, 2), None, None, image_folder, 2, 2, enable_text=False, enable_image=True)
        batches = list(r)
        got.append([b["image_tensor"].shape[0] for b in batches])
        assert all(tuple(b["image_tensor"].shape[1:]) == (3, 224, 224) for b in batches)
        assert all(str(b["image_tensor"].dtype) == "torch.float32" for b in batches)
    assert got == [[2, 2], [2, 1]]


def test_webdataset_reader_batch_shapes(tar_shards):
    """Reference expectation: 11 images in 4 tars, 2 partitions, batch 2 -> [[2,2,2],[2,2,1]] (test_reader.py:60-61)."""
    from clip_retrieval_amd.reader import HashTokenizer, WebdatasetReader
    from clip_retrieval_amd.runner import Sampler

    got = []
    for pid in range(2):
        r = WebdatasetReader(Sampler(pid, 2), None, HashTokenizer(), tar_shards, 2, 2, enable_text=True, enable_image=True)
        batches = list(r)
        got.append([b["image_tensor"].shape[0] for b in batches])
        for b in batches:
            assert b["text_tokens"].shape[1] == 77 and len(b["text"]) == b["image_tensor"].shape[0]
            assert int(b["text_tokens"][0].max()) == 49407  # EOT is the highest id
    assert got == [[2, 2, 2], [2, 2, 1]]


def test_reader_skips_corrupt_image(tmp_path, image_folder):
    from clip_retrieval_amd.reader import FilesReader
    from clip_retrieval_amd.runner import Sampler

    (tmp_path / "imgs" / "000_bad.jpg").write_bytes(b"not a jpeg")
    r = FilesReader(Sampler(0, 1), None, None, image_folder, 4, 1, enable_text=False, enable_image=True)
    assert sum(b["image_tensor"].shape[0] for b in r) == 7


def test_runner_with_stub_mapper_writes_reference_layout(tmp_path, image_folder):
    """Runner + reader + writer plumbing with a stub mapper (the real one needs the GPU):
    partition 0 of 2 over 7 images -> img_emb_0.npy with 4 rows (reference test_runner.py:76-77)."""
    from clip_retrieval_amd.reader import FilesReader
    from clip_retrieval_amd.runner import NullLogger, Runner
    from clip_retrieval_amd.writer import NumpyWriter

    class StubMapper:
        def __call__(self, item):
            n = item["image_tensor"].shape[0]
            return {"image_embs": np.ones((n, 8), np.float16), "text_embs": None,
                    "image_filename": item["image_filename"], "text": None, "metadata": None}

    out = tmp_path / "out"
    logs = {}

    def logger_builder(i):
        logs[i] = NullLogger(i)
        return logs[i]

    runner = Runner(
        reader_builder=lambda s: FilesReader(s, None, None, image_folder, 2, 1, enable_text=False, enable_image=True),
        mapper_builder=StubMapper,
        writer_builder=lambda i: NumpyWriter(i, str(out), False, True, False, 2),
        logger_builder=logger_builder,
        output_partition_count=2,
    )
    runner(0)
    runner(1)
    assert np.load(out / "img_emb" / "img_emb_0.npy").shape == (4, 8)
    assert np.load(out / "img_emb" / "img_emb_1.npy").shape == (3, 8)
    keys = {"start_time", "end_time", "read_duration", "inference_duration", "write_duration", "total_duration", "sample_count"}
    assert all(set(r) == keys for r in logs[0].records) and sum(r["sample_count"] for r in logs[0].records) == 4


def test_reader_runner_writer_match_reference_run(tmp_path):
    """PIN of the plumbing on both sides of the mapper: tests/golden/reference_reader_runner.json was recorded by running
    the reference's own FilesReader + Runner + NumpyWriter (tests/golden/make_golden_reader.py) over the deterministic
    folder of tests/golden/reader_fixture.py.  Ours must yield the same batches (order, file names, bit-identical image
    tensors, the corrupt image skipped) and write byte-identical .npy files and identical parquet rows."""
    import hashlib
    import sys

    import pandas as pd
    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from reader_fixture import ListLogger, StubMapper, make_folder

    from clip_retrieval_amd.reader import FilesReader, HashTokenizer, clip_preprocess
    from clip_retrieval_amd.runner import Runner
    from clip_retrieval_amd.writer import NumpyWriter

    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_reader_runner.json")) as f:
        golden = json.load(f)
    folder = make_folder(str(tmp_path / "in"))
    out = str(tmp_path / "out")
    seen = {}
    preprocess = lambda im: torch.from_numpy(clip_preprocess(im))  # noqa: E731

    class Recording:
        def __init__(self, sampler):
            self.inner = FilesReader(sampler, preprocess, HashTokenizer(), folder, 3, 0, enable_text=False, enable_image=True,
                                     enable_metadata=False)
            self.pid = sampler.output_partition_id

        def __iter__(self):
            for b in self.inner:
                seen.setdefault(self.pid, []).append({
                    "image_filename": [os.path.basename(p) for p in b["image_filename"]], "keys": sorted(b.keys()),
                    "image_shape": list(b["image_tensor"].shape), "image_dtype": str(b["image_tensor"].dtype),
                    "image_sha256": hashlib.sha256(b["image_tensor"].numpy().tobytes()).hexdigest()})
                yield b

    r = Runner(reader_builder=Recording, mapper_builder=StubMapper,
               writer_builder=lambda i: NumpyWriter(i, out, False, True, False, 2), logger_builder=ListLogger,
               output_partition_count=2)
    for i in range(2):
        r(i)
    assert [seen.get(i, []) for i in range(2)] == golden["partitions"]
    for rel, want in golden["files"].items():
        p = os.path.join(out, rel)
        assert os.path.exists(p), rel
        if rel.endswith(".npy"):
            assert hashlib.sha256(open(p, "rb").read()).hexdigest() == want, rel
        else:
            df = pd.read_parquet(p)
            got_rows = json.loads(df.to_json(orient="records"))
            # the reference stores absolute paths of its temporary folder: compare base names
            strip = lambda rows: [{k: (os.path.basename(v) if k == "image_path" else v) for k, v in row.items()} for row in rows]  # noqa: E731
            assert list(df.columns) == want["columns"] and strip(got_rows) == strip(want["rows"]), rel


def test_committed_bench_line_follows_the_contract():
    """The newest bench line committed under profiles/ carries every field the driver's contract names (bench.py is only
    runnable on the GPU box; this guards the JSON schema on the CPU)."""
    import glob
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logs = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench.log")))
    assert logs, "no bench log committed under profiles/"
    line = [l for l in open(logs[-1]) if l.startswith('{"metric"')][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    k = d["knn"]["roofline"]
    assert k["bound"] == "hbm" and abs(k["frac"] - k["achieved"] / k["peak"]) < 1e-3
    assert d["parity"]["ok"] is True


# ---------------------------------------------------------------------------------------------------------------
# round 2: pipelined Runner, LoggerWriter, lazy reader, the reference's key rule, worker task split, preprocess
# ---------------------------------------------------------------------------------------------------------------
class _AsyncFakeMapper:
    """submit/collect mapper (the shape of ours) over a deterministic "embedding": records the call order."""

    def __init__(self, log):
        self.log = log

    def _emb(self, item):
        x = item["image_tensor"].numpy()
        return x.reshape(x.shape[0], -1)[:, :8].astype(np.float16)

    def submit(self, item):
        self.log.append(("submit", item["image_filename"][0]))
        return {"item": item}

    def collect(self, h):
        item = h["item"]
        self.log.append(("collect", item["image_filename"][0]))
        return {"image_embs": self._emb(item), "text_embs": None, "image_filename": item["image_filename"], "text": None, "metadata": None}

    def __call__(self, item):
        return self.collect({"item": item})


def test_pipelined_runner_equals_the_serial_loop_and_overlaps(tmp_path, image_folder):
    """A mapper with submit/collect is driven one batch ahead (batch n+1 submitted before batch n is collected); files,
    row order and the seven stat keys are those of the serial reference loop."""
    from clip_retrieval_amd.reader import FilesReader, clip_preprocess
    from clip_retrieval_amd.runner import LoggerWriter, NullLogger, Runner
    from clip_retrieval_amd.writer import NumpyWriter

    outs, logs = {}, {}
    for mode in ("pipelined", "serial"):
        log = []
        mapper = _AsyncFakeMapper(log)
        if mode == "serial":
            mapper = mapper.__call__  # a plain callable: the reference's loop
        out = tmp_path / mode
        loggers = []

        def logger_builder(i, out=out, loggers=loggers):
            lg = LoggerWriter(i, str(out / "stats"))
            loggers.append(lg)
            return lg

        r = Runner(lambda s: FilesReader(s, clip_preprocess, None, str(image_folder), 2, 2, enable_text=False),
                   lambda mapper=mapper: mapper,
                   lambda i, out=out: NumpyWriter(i, str(out), False, True, False, 1), logger_builder, 1)
        r(0)
        outs[mode] = np.load(out / "img_emb" / "img_emb_0.npy")
        logs[mode] = log
        stats = json.load(open(out / "stats" / "0.json"))
        assert set(stats) == {"start_time", "end_time", "read_duration", "inference_duration", "write_duration", "total_duration", "sample_count"}
        assert stats["sample_count"] == outs[mode].shape[0]
        assert not (out / "stats" / "wip_0.json").exists()
    assert np.array_equal(outs["pipelined"], outs["serial"]) and outs["serial"].shape[0] == 7
    order = logs["pipelined"]
    assert [k for k, _ in order[:3]] == ["submit", "submit", "collect"], order  # batch 1 submitted before batch 0 is collected
    assert [n for k, n in order if k == "collect"] == [n for k, n in order if k == "submit"]  # collected in submission order


def test_prefetcher_propagates_reader_errors():
    from clip_retrieval_amd.runner import _Prefetcher

    def gen():
        yield 1
        raise ValueError("decode exploded")

    p = _Prefetcher(gen(), 2)
    assert next(p) == 1
    with pytest.raises(ValueError):
        next(p)


def test_reader_reads_lazily(image_folder):
    """At most workers * 2 + batch_size samples are in flight: the bytes of later files are not read before the consumer
    asks (ThreadPoolExecutor.map would have read and decoded the whole partition up front)."""
    from clip_retrieval_amd.reader import FilesReader, clip_preprocess
    from clip_retrieval_amd.runner import Sampler

    r = FilesReader(Sampler(0, 1), clip_preprocess, None, str(image_folder), 1, 1, enable_text=False)
    reads = []
    inner = r._raw_samples

    def counting():
        for raw in inner():
            reads.append(raw["key"])
            yield raw

    r._raw_samples = counting
    it = iter(r)
    next(it)
    assert len(reads) <= 1 * 2 + 1 + 1, f"{len(reads)} of 7 files were read before the first batch was consumed"
    assert sum(b["image_tensor"].shape[0] for b in it) == 6


def test_folder_keys_follow_the_reference_rule(tmp_path):
    """Single modality: full relative path INCLUDING the extension, string order (reference reader.py:10-51): a.jpg and
    a.png are two samples and 'a-b.jpg' sorts before 'a.png'.  Several modalities are paired by stem, order of the first
    modality, stem collisions reported."""
    from clip_retrieval_amd.reader import folder_to_keys

    for name in ("a.png", "a-b.jpg", "a.jpg", "sub/c.JPG", "a.txt", "a-b.txt", "zz.txt"):
        p = tmp_path / name
        p.parent.mkdir(exist_ok=True)
        p.write_bytes(b"x")
    keys, tf, imf, mf = folder_to_keys(str(tmp_path), enable_text=False, enable_image=True)
    assert keys == ["a-b.jpg", "a.jpg", "a.png", "sub/c.JPG"] and tf is None and mf is None
    assert all(k in imf for k in keys)
    keys, tf, imf, _ = folder_to_keys(str(tmp_path), enable_text=True, enable_image=True)
    assert keys == ["a-b", "a"]  # order of the text files 'a-b.txt' < 'a.txt'; zz has no image, sub/c no caption
    assert tf["a"].name == "a.txt" and imf["a"].name == "a.jpg" and imf["a-b"].name == "a-b.jpg"


def test_worker_deals_partitions_to_gpus(monkeypatch):
    """gpu_worker = slurm_worker.py:40-61 with the launcher's variables: rank r of 8 gets get_task_list(NUM_TASKS, 8, r)
    and binds to device LOCAL_RANK; brace patterns expand like the braceexpand package on the reference's shard specs."""
    import clip_retrieval_amd.worker as W

    assert W.braceexpand("/d/{000..002}.tar") == ["/d/000.tar", "/d/001.tar", "/d/002.tar"]
    assert W.braceexpand("s3://b/{8..10}_{a,b}.tar") == ["s3://b/8_a.tar", "s3://b/8_b.tar", "s3://b/9_a.tar", "s3://b/9_b.tar", "s3://b/10_a.tar", "s3://b/10_b.tar"]
    seen = {}
    monkeypatch.setattr(W, "worker", lambda tasks, **kw: seen.update(tasks=tasks, **kw))
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "2")
    monkeypatch.setenv("LOCAL_RANK", "2")
    W.gpu_worker(input_dataset="x", output_folder="y", output_partition_count=21)
    assert seen["tasks"] == W.get_task_list(21, 8, 2) == [6, 7, 8] and seen["device"] == 2
    covered = sorted(t for r in range(8) for t in W.get_task_list(21, 8, r))
    assert covered == list(range(21))


def test_clip_preprocess_geometry_and_values():
    """CLIP's transform restated: Resize(shorter side -> S, bicubic, torchvision's int(S * long / short) for the long side)
    -> CenterCrop(S) (offsets int(round((dim - S) / 2))) -> RGB -> /255 -> (x - mean) / std, channel-first float32."""
    from PIL import Image

    from clip_retrieval_amd.reader import CLIP_MEAN, CLIP_STD, clip_preprocess

    S = 224
    for (w, h) in [(640, 480), (123, 456), (224, 224), (225, 224), (1001, 333)]:
        img = Image.fromarray(np.random.default_rng(w).integers(0, 255, (h, w, 3), dtype=np.uint8))
        out = clip_preprocess(img, S)
        assert out.shape == (3, S, S) and out.dtype == np.float32
        nw, nh = (S, int(S * h / w)) if w <= h else (int(S * w / h), S)
        ref = img.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - S) / 2.0)), int(round((nh - S) / 2.0))
        ref = np.asarray(ref.crop((left, top, left + S, top + S)).convert("RGB"), dtype=np.float32) / np.float32(255)
        ref = ((ref - CLIP_MEAN) / CLIP_STD).transpose(2, 0, 1)
        assert np.allclose(out, ref, atol=1e-6)
    # a constant grey image maps to (v/255 - mean)/std in every pixel; palette / greyscale inputs are converted to RGB
    g = clip_preprocess(Image.new("L", (300, 200), 128), S)
    assert np.allclose(g[:, 0, 0], (128 / 255 - CLIP_MEAN) / CLIP_STD, atol=1e-6) and np.allclose(g, g[:, :1, :1], atol=1e-6)


def test_u8_preprocess_is_the_float_transform_before_normalisation(image_folder):
    """clip_preprocess_u8 (resize + crop on the host, normalisation on the GPU) yields exactly the uint8 pixels the float
    transform normalises, and a reader built with it collates uint8 [B, S, S, 3] batches."""
    from PIL import Image

    from clip_retrieval_amd.reader import CLIP_MEAN, CLIP_STD, FilesReader, clip_preprocess, clip_preprocess_u8
    from clip_retrieval_amd.runner import Sampler

    img = Image.fromarray(np.random.default_rng(5).integers(0, 255, (300, 411, 3), dtype=np.uint8))
    u8 = clip_preprocess_u8(img, 224)
    assert u8.shape == (224, 224, 3) and u8.dtype == np.uint8
    want = ((u8.astype(np.float32) / np.float32(255.0) - CLIP_MEAN) / CLIP_STD).transpose(2, 0, 1)  # torchvision's arithmetic
    assert np.array_equal(clip_preprocess(img, 224), want)
    r = FilesReader(Sampler(0, 1), clip_preprocess_u8, None, str(image_folder), 4, 2, enable_text=False)
    b = next(iter(r))
    assert b["image_tensor"].dtype.__str__() == "torch.uint8" and tuple(b["image_tensor"].shape) == (4, 224, 224, 3)


def test_decode_processes_equal_the_thread_pool(tmp_path, tar_shards):
    """num_prepro_workers > 1 decodes in worker PROCESSES (the reference's DataLoader workers, reader.py:184-205): same
    batches, bit for bit and in the same order, as decoding inside this process; a corrupt member is skipped the same way;
    a transform that cannot be pickled falls back to threads."""
    import torch

    from clip_retrieval_amd.reader import HashTokenizer, WebdatasetReader, _DecodePool, clip_preprocess, clip_preprocess_u8
    from clip_retrieval_amd.runner import Sampler

    bad = tmp_path / "bad.tar"
    with tarfile.open(bad, "w") as tf:
        for name, data in (("x0.jpg", b"not a jpeg"), ("x0.txt", b"broken"), ("x1.jpg", _jpeg(120, 90, 5)), ("x1.txt", b"fine")):
            ti = tarfile.TarInfo(name)
            ti.size = len(data)
            tf.addfile(ti, io.BytesIO(data))
    shards = tar_shards + [str(bad)]
    for prep in (clip_preprocess, clip_preprocess_u8):
        out = {}
        for procs in (False, True):
            r = WebdatasetReader(Sampler(0, 1), prep, HashTokenizer(), shards, 5, 3)
            r.use_processes, r.chunk = procs, 2
            out[procs] = list(r)
        # the reference's loader order with 3 workers (round 4, WebdatasetReader.reference_batch_order): shards 0, 3 -> stream 0
        # (4 + 2 samples: batches 5, 1), shards 1, bad -> stream 1 (3 + 1: batch 4), shard 2 -> stream 2 (batch 2), round-robin
        assert [b["image_tensor"].shape[0] for b in out[True]] == [5, 4, 2, 1]
        assert [b["image_filename"] for b in out[True]][1] == ["00004", "00005", "00006", "x1"]
        for a, b in zip(out[False], out[True]):
            assert torch.equal(a["image_tensor"], b["image_tensor"]) and a["image_tensor"].dtype == b["image_tensor"].dtype
            assert torch.equal(a["text_tokens"], b["text_tokens"]) and a["text"] == b["text"]
            assert a["image_filename"] == b["image_filename"]
    assert any(p is not None for p in _DecodePool._pools.values()), "the process path did not start any worker"
    n_pools = len(_DecodePool._pools)
    r = WebdatasetReader(Sampler(0, 1), lambda im: clip_preprocess(im), HashTokenizer(), tar_shards, 4, 3)  # noqa: E731
    assert sum(b["image_tensor"].shape[0] for b in r) == 11 and len(_DecodePool._pools) == n_pools
    _DecodePool.shutdown()


def test_config1_plumbing_with_the_oracle_mapper(tmp_path):
    """SURVEY config 1 (the reference's own CPU-runnable case, scaled down): synthetic JPEGs + captions in webdataset-style
    tars -> WebdatasetReader (decode processes) -> Runner -> a mapper with ClipMapper's call shape whose model is the fp32
    ORACLE (no GPU here) -> NumpyWriter + LoggerWriter.  Asserts the reference's output layout (writer.py:67-106): file names,
    dtypes, shapes, row counts per partition, unit-norm fp16 rows, and that row i of the embeddings belongs to row i of the
    metadata."""
    import pandas as pd
    from PIL import Image

    from clip_retrieval_amd.reader import HashTokenizer, WebdatasetReader, clip_preprocess
    from clip_retrieval_amd.runner import LoggerWriter, Runner
    from clip_retrieval_amd.writer import NumpyWriter
    from oracle.clip_oracle import ARCHS, HFClipOracle, mapper_semantics

    arch = ARCHS["tiny-B/32"]
    oracle = HFClipOracle(arch, seed=0)
    rng = np.random.default_rng(0)
    shards, k = [], 0
    for t in range(4):
        p = tmp_path / f"{t:03d}.tar"
        with tarfile.open(p, "w") as tf:
            for _ in range(25):
                g = np.linspace(0, 255, 256, dtype=np.float32)
                img = (g[None, :, None] * 0.5 + g[:, None, None] * 0.5 + rng.normal(0, 8, (256, 256, 3))).clip(0, 255).astype(np.uint8)
                buf = io.BytesIO()
                Image.fromarray(img).save(buf, format="JPEG", quality=90)
                for ext, data in (("jpg", buf.getvalue()), ("txt", f"caption {k}".encode())):
                    ti = tarfile.TarInfo(f"{k:06d}.{ext}")
                    ti.size = len(data)
                    tf.addfile(ti, io.BytesIO(data))
                k += 1
        shards.append(str(p))

    class OracleMapper:  # ClipMapper.__call__'s contract (mapper.py:49-78) on the CPU oracle
        def __call__(self, item):
            img16, _ = mapper_semantics(oracle.encode_image(item["image_tensor"]))
            txt16, _ = mapper_semantics(oracle.encode_text(item["text_tokens"].clamp(max=arch.vocab - 1)))
            return {"image_embs": img16, "text_embs": txt16, "image_filename": item["image_filename"], "text": item["text"],
                    "metadata": None}

    out = tmp_path / "out"
    tok = HashTokenizer(arch.ctx_len, arch.vocab)
    runner = Runner(
        reader_builder=lambda s: WebdatasetReader(s, functools.partial(clip_preprocess, size=arch.image_size), tok, shards, 16, 2),
        mapper_builder=OracleMapper,
        writer_builder=lambda i: NumpyWriter(i, str(out), True, True, False, 2),
        logger_builder=lambda i: LoggerWriter(i, str(out / "stats")),
        output_partition_count=2,
    )
    runner(0)
    runner(1)
    total = 0
    for i in range(2):
        img = np.load(out / "img_emb" / f"img_emb_{i}.npy")
        txt = np.load(out / "text_emb" / f"text_emb_{i}.npy")
        meta = pd.read_parquet(out / "metadata" / f"metadata_{i}.parquet")
        assert img.dtype == np.float16 and txt.dtype == np.float16 and img.shape == (50, arch.embed_dim) == txt.shape
        assert len(meta) == 50 and list(meta.columns)[:2] == ["image_path", "caption"]
        assert np.allclose(np.linalg.norm(img.astype(np.float32), axis=1), 1, atol=2e-3)
        # partition i holds shards i, i + 2 (Sampler: every count-th shard), 25 samples each; rows come in the reference loader's
        # order for its 2 workers (reader.py:184-205): each worker batches its own shard, batches alternate -- 16 of shard i,
        # 16 of shard i + 2, then the 9 left of each
        a, b = [f"caption {j}" for j in range(25 * i, 25 * i + 25)], [f"caption {j}" for j in range(25 * (i + 2), 25 * (i + 2) + 25)]
        want = a[:16] + b[:16] + a[16:] + b[16:]
        assert list(meta["caption"]) == want
        total += len(meta)
        st = json.loads((out / "stats" / f"{i}.json").read_text())  # LoggerWriter: summed stats of the partition (logger.py:13-62)
        assert st["sample_count"] == 50 and not (out / "stats" / f"wip_{i}.json").exists()
    assert total == 100


def test_pipelined_runner_drops_in_flight_tickets_when_a_batch_fails(tmp_path, image_folder):
    """if collect / the writer raises, the batch that was already submitted must still be waited for (discarded),
    otherwise its staging slot of the long-lived encoder is never released."""
    from clip_retrieval_amd.reader import FilesReader, clip_preprocess
    from clip_retrieval_amd.runner import NullLogger, Runner

    class Mapper(_AsyncFakeMapper):
        def __init__(self, log):
            super().__init__(log)
            self.outstanding = set()

        def submit(self, item):
            h = super().submit(item)
            self.outstanding.add(id(h))
            return h

        def collect(self, h):
            self.outstanding.discard(id(h))
            return super().collect(h)

        def discard(self, h):
            self.log.append(("discard", h["item"]["image_filename"][0]))
            self.outstanding.discard(id(h))

    class BadWriter:
        def __init__(self):
            self.n = 0

        def __call__(self, emb):
            self.n += 1
            if self.n == 2:
                raise RuntimeError("disk full")

        def flush(self):
            pass

    log = []
    m = Mapper(log)
    r = Runner(lambda s: FilesReader(s, clip_preprocess, None, str(image_folder), 2, 2, enable_text=False), lambda: m,
               lambda i: BadWriter(), lambda i: NullLogger(i), 1)
    with pytest.raises(RuntimeError, match="disk full"):
        r(0)
    assert not m.outstanding, f"tickets left in flight: {log}"
    assert any(k == "discard" for k, _ in log)


def test_folder_rows_is_the_concatenation_of_the_partitions(tmp_path):
    """knn.FolderRows (the row source of the streamed IVF build / load): any [lo, hi) of the virtual matrix, chunks across file
    borders, sorted gathers, and the manifest that save_index records (file names + row counts: ids are row numbers)."""
    from clip_retrieval_amd.knn import FolderRows

    rng = np.random.default_rng(0)
    parts = [rng.standard_normal((n, 16)).astype(np.float16) for n in (5, 0, 7, 3)]
    for i, a in enumerate(parts):
        np.save(tmp_path / f"img_emb_{i}.npy", a)
    allr = np.concatenate(parts)
    fr = FolderRows(str(tmp_path))
    assert fr.n == 15 and fr.d == 16 and fr.starts.tolist() == [0, 5, 5, 12, 15]
    for lo in range(16):
        for hi in range(lo, 16):
            got = fr.rows(lo, hi)
            assert got.dtype == np.float16 and got.flags["C_CONTIGUOUS"] and np.array_equal(got, allr[lo:hi])
    assert [(o, len(x)) for o, x in fr.chunks(2, 15, 4)] == [(2, 4), (6, 4), (10, 4), (14, 1)]
    idx = np.array([0, 4, 5, 11, 12, 14])
    assert np.array_equal(fr.take(idx), allr[idx])
    assert fr.manifest()["files"] == [["img_emb_0.npy", 5], ["img_emb_1.npy", 0], ["img_emb_2.npy", 7], ["img_emb_3.npy", 3]]
    with pytest.raises(IndexError):
        fr.rows(3, 16)
    np.save(tmp_path / "img_emb_4.npy", np.zeros((2, 8), np.float16))
    with pytest.raises(ValueError):
        FolderRows(str(tmp_path))


def test_kmeans_rebalancing_splits_the_big_lists_and_retires_the_small_ones():
    """knn.rebalance_centroids (the split-and-merge step of the IVF k-means, reference: autofaiss / faiss Clustering behind
    clip_index.py:12-66).  Eight heavy clusters and fifty-six light ones, one initial centroid on each: plain Lloyd keeps one list per
    cluster (3 000 rows against 50); with the step the light lists' centroids are handed to the heavy clusters, whose lists end
    several times smaller -- the same number of lists, every row still in exactly one.  (The kernels' k-means is checked on the GPU:
    tests/test_knn_gpu.py.)"""
    from clip_retrieval_amd.knn import rebalance_centroids

    rng = np.random.default_rng(5)
    d, nlist = 32, 64
    pops = np.r_[np.full(8, 3000), np.full(56, 50)]
    centres = rng.standard_normal((nlist, d))
    x = np.concatenate([c + 0.35 * rng.standard_normal((n, d)) for c, n in zip(centres, pops)])
    x /= np.linalg.norm(x, axis=1, keepdims=True)

    def lloyd(balance):
        r = np.random.default_rng(7)
        cent = (centres / np.linalg.norm(centres, axis=1, keepdims=True)).astype(np.float32)
        splits = 0
        for it in range(10):
            a = np.argmax(x @ cent.T, axis=1)
            sizes = np.bincount(a, minlength=nlist)
            for l in np.flatnonzero(sizes):
                m = x[a == l].mean(axis=0)
                cent[l] = m / np.linalg.norm(m)
            if balance and it < 8:
                splits += rebalance_centroids(cent, sizes, r)
        return np.bincount(np.argmax(x @ cent.T, axis=1), minlength=nlist), splits

    (plain, _), (bal, splits) = lloyd(False), lloyd(True)
    assert plain.sum() == bal.sum() == len(x) and len(bal) == nlist
    assert plain.max() >= 2900 and splits >= 16
    assert bal.max() <= plain.max() / 2.5, (plain.max(), bal.max())
    assert bal.max() <= 2.2 * bal.mean(), (bal.max(), bal.mean())    # nothing left above the split threshold (2 x the mean) by much


def test_embedding_files_ignore_the_sidecars_of_a_saved_index(tmp_path):
    """save_index() writes ivf_centroids.npy / ivf_lists.npy; saved INTO the embeddings folder they must not become two more
    partitions at the next load (knn.embedding_files is what load_index / FolderRows glob with)."""
    from clip_retrieval_amd.knn import embedding_files

    for name in ("img_emb_0.npy", "img_emb_1.npy", "ivf_centroids.npy", "ivf_lists.npy"):
        np.save(tmp_path / name, np.zeros((2, 4), np.float16))
    got = [os.path.basename(f) for f in embedding_files(str(tmp_path))]
    assert got == ["img_emb_0.npy", "img_emb_1.npy"]
