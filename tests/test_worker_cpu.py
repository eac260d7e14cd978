"""CPU tests of the callers around the encode seam that round 2 left untested:
the readers on the REFERENCE's own fixtures, worker.gpu_worker as two real processes (one per "GPU") with an oracle-backed
mapper, the args-file precedence of the CLI, and SURVEY config 1 at its stated size.  No GPU: the mapper is the fp32 oracle
(test infrastructure), patched in where worker.worker() would build the HIP ClipMapper."""
import io
import json
import os
import subprocess
import sys
import tarfile
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FIX = "/root/reference/tests/test_clip_inference"


@pytest.mark.skipif(not os.path.isdir(REF_FIX), reason="the reference checkout (build container only) holds these fixtures")
def test_readers_on_the_reference_fixture_files():
    """The reference's own reader test (tests/test_clip_inference/test_reader.py:18-61) on the reference's own files:
    test_images (7 JPEGs) -> [[2, 2], [2, 1]], test_tars (4 tars, 11 samples) -> [[2, 2, 2], [2, 2, 1]] for 2 partitions of
    batch size 2; tensors are f32 [n, 3, 224, 224] like the reference's `image_tensor`."""
    import torch

    from clip_retrieval_amd.reader import FilesReader, WebdatasetReader, clip_preprocess
    from clip_retrieval_amd.runner import Sampler

    tars = [f"{REF_FIX}/test_tars/image{i + 1}.tar" for i in range(4)]
    got = {"files": [], "webdataset": []}
    for pid in range(2):
        s = Sampler(pid, 2)
        r = FilesReader(s, clip_preprocess, None, f"{REF_FIX}/test_images", 2, 2, enable_text=False, enable_image=True, enable_metadata=False)
        batches = list(r)
        got["files"].append([b["image_tensor"].shape[0] for b in batches])
        assert all(b["image_tensor"].dtype == torch.float32 and tuple(b["image_tensor"].shape[1:]) == (3, 224, 224) for b in batches)
        r = WebdatasetReader(s, clip_preprocess, None, tars, 2, 2, enable_text=False, enable_image=True, enable_metadata=False)
        got["webdataset"].append([b["image_tensor"].shape[0] for b in r])
    assert got["files"] == [[2, 2], [2, 1]], got          # test_reader.py:58-59
    assert got["webdataset"] == [[2, 2, 2], [2, 2, 1]], got  # test_reader.py:60-61


def _write_shards(folder, n_shards, per_shard, size=64):
    from PIL import Image

    rng = np.random.default_rng(0)
    paths, k = [], 0
    for t in range(n_shards):
        p = os.path.join(folder, f"{t:03d}.tar")
        with tarfile.open(p, "w") as tf:
            for _ in range(per_shard):
                g = np.linspace(0, 255, size, dtype=np.float32)
                img = (g[None, :, None] * 0.5 + g[:, None, None] * 0.5 + rng.normal(0, 8, (size, size, 3))).clip(0, 255).astype(np.uint8)
                buf = io.BytesIO()
                Image.fromarray(img).save(buf, format="JPEG", quality=90)
                for ext, data in (("jpg", buf.getvalue()), ("txt", f"caption {k}".encode())):
                    ti = tarfile.TarInfo(f"{k:06d}.{ext}")
                    ti.size = len(data)
                    tf.addfile(ti, io.BytesIO(data))
                k += 1
        paths.append(p)
    return paths


# the child process: worker.gpu_worker exactly as a launcher starts it, with the two HIP-backed builders of worker.worker()
# (load_clip for the reader, ClipMapper) replaced by oracle-backed stand-ins of the same call shape
_CHILD = textwrap.dedent("""
    import functools, json, os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import clip_retrieval_amd.worker as W
    from clip_retrieval_amd.reader import HashTokenizer, clip_preprocess
    from oracle.clip_oracle import ARCHS, HFClipOracle, mapper_semantics

    arch = ARCHS["tiny-B/32"]
    oracle = HFClipOracle(arch, seed=0)

    class _Enc:  # what worker.reader_builder reads off the model
        pass
    _Enc.arch = arch

    class _Model:
        _enc = _Enc

    def load_clip(**kw):
        return _Model, functools.partial(clip_preprocess, size=arch.image_size), HashTokenizer(arch.ctx_len, arch.vocab)

    class OracleMapper:  # ClipMapper's constructor and call contract (mapper.py:19-78) on the CPU oracle
        def __init__(self, enable_image, enable_text, enable_metadata, use_mclip, clip_model, use_jit, mclip_model,
                     warmup_batch_size=1, clip_cache_path=None, device=None):
            self.device = device
        def __call__(self, item):
            x = item["image_tensor"]
            if x.dtype != np.float32 and str(x.dtype) != "torch.float32":  # gpu_normalise=True hands uint8 NHWC pixels over
                raise SystemExit("oracle mapper wants the float image_tensor")
            img16, _ = mapper_semantics(oracle.encode_image(x))
            txt16, _ = mapper_semantics(oracle.encode_text(item["text_tokens"].clamp(max=arch.vocab - 1)))
            return {{"image_embs": img16, "text_embs": txt16, "image_filename": item["image_filename"], "text": item["text"], "metadata": None}}

    W.load_clip = load_clip
    W.ClipMapper = OracleMapper
    W._main({argv!r})
""")


def test_gpu_worker_as_two_processes_writes_disjoint_partitions(tmp_path):
    """slurm_worker.py:40-61 / worker.py:22-127 end to end on the CPU: two processes (RANK 0 / 1 of WORLD_SIZE 2), five output
    partitions over five tar shards, the worker arguments coming from a WORKER_ARGS_PATH json (as the reference's slurm
    distributor writes it) with one of them overridden on the command line.  Every partition must be written exactly once, by
    the rank get_task_list gives it to, with the rows of its own shard in order."""
    import pandas as pd

    from clip_retrieval_amd.runner import get_task_list

    shards = _write_shards(str(tmp_path), 5, 6)
    out = tmp_path / "out"
    args = {"input_dataset": shards, "output_folder": str(out), "output_partition_count": 5, "input_format": "webdataset",
            "batch_size": 64, "num_prepro_workers": 2, "enable_text": True, "enable_image": True, "enable_metadata": False,
            "clip_model": "tiny-B/32", "gpu_normalise": False}
    args_path = tmp_path / "worker_args.json"
    args_path.write_text(json.dumps(args))
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), WORKER_ARGS_PATH=str(args_path),
                   CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
        code = _CHILD.format(root=ROOT, argv=["--batch_size", "4"])  # the command line overrides the file's 64
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    for rank in range(2):
        assert f"processing tasks {get_task_list(5, 2, rank)}" in logs[rank], logs[rank]
    assert sorted(os.listdir(out / "img_emb")) == [f"img_emb_{i}.npy" for i in range(5)]
    for i in range(5):
        img = np.load(out / "img_emb" / f"img_emb_{i}.npy")
        txt = np.load(out / "text_emb" / f"text_emb_{i}.npy")
        meta = pd.read_parquet(out / "metadata" / f"metadata_{i}.parquet")
        assert img.shape == (6, 512) == txt.shape and img.dtype == np.float16  # tiny-B/32: embed_dim 512
        assert list(meta["caption"]) == [f"caption {j}" for j in range(6 * i, 6 * i + 6)]  # partition i = shard i (Sampler)
        st = json.loads((out / "stats" / f"{i}.json").read_text())
        assert st["sample_count"] == 6
    # batch_size 4 from the command line, not the file's 64: two batches per partition
    assert sum("Starting work on task" in line for line in "\n".join(logs).splitlines()) == 5


def test_gpu_worker_argument_precedence(monkeypatch, tmp_path):
    """options the CLI leaves unset must come from the WORKER_ARGS_PATH file, explicit ones win."""
    import clip_retrieval_amd.worker as W

    seen = {}
    monkeypatch.setattr(W, "worker", lambda tasks, **kw: seen.update(tasks=tasks, **kw))
    p = tmp_path / "a.json"
    p.write_text(json.dumps({"input_dataset": "in", "output_folder": "of", "output_partition_count": 3, "batch_size": 77,
                             "enable_text": False, "input_format": "webdataset"}))
    monkeypatch.setenv("WORKER_ARGS_PATH", str(p))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    W._main(["--output_folder", "cli_folder", "--enable_image", "false"])
    assert seen["batch_size"] == 77 and seen["enable_text"] is False and seen["input_format"] == "webdataset"
    assert seen["output_folder"] == "cli_folder" and seen["enable_image"] is False and seen["tasks"] == [0, 1, 2]
    monkeypatch.delenv("WORKER_ARGS_PATH")
    with pytest.raises(ValueError, match="output_partition_count"):
        W._main(["--input_dataset", "x", "--output_folder", "y"])


def test_config1_at_its_stated_size(tmp_path):
    """SURVEY 8(d) config 1 as written: ViT-B/32 at FULL depth, 1 000 synthetic 256 x 256 JPEGs with 1 000 captions in ONE
    webdataset-style tar, through reader -> Runner -> a ClipMapper-shaped mapper on the fp32 oracle -> NumpyWriter, no GPU.
    Asserts the reference's output layout (writer.py:67-106) and the row count 1 000."""
    import functools

    import pandas as pd
    import torch

    from clip_retrieval_amd.reader import HashTokenizer, WebdatasetReader, clip_preprocess
    from clip_retrieval_amd.runner import LoggerWriter, Runner
    from clip_retrieval_amd.writer import NumpyWriter
    from oracle.clip_oracle import ARCHS, HFClipOracle, mapper_semantics

    arch = ARCHS["ViT-B/32"]
    oracle = HFClipOracle(arch, seed=0, threads=os.cpu_count() or 1)
    shard = _write_shards(str(tmp_path), 1, 1000, size=256)

    class OracleMapper:
        def __call__(self, item):
            img16, _ = mapper_semantics(oracle.encode_image(item["image_tensor"]))
            txt16, _ = mapper_semantics(oracle.encode_text(item["text_tokens"].clamp(max=arch.vocab - 1)))
            return {"image_embs": img16, "text_embs": txt16, "image_filename": item["image_filename"], "text": item["text"], "metadata": None}

    out = tmp_path / "out"
    tok = HashTokenizer(arch.ctx_len, arch.vocab)
    Runner(reader_builder=lambda s: WebdatasetReader(s, functools.partial(clip_preprocess, size=arch.image_size), tok, shard, 100, 4),
           mapper_builder=OracleMapper, writer_builder=lambda i: NumpyWriter(i, str(out), True, True, False, 1),
           logger_builder=lambda i: LoggerWriter(i, str(out / "stats")), output_partition_count=1)(0)
    img = np.load(out / "img_emb" / "img_emb_0.npy")
    txt = np.load(out / "text_emb" / "text_emb_0.npy")
    meta = pd.read_parquet(out / "metadata" / "metadata_0.parquet")
    assert img.shape == (1000, 512) == txt.shape and img.dtype == np.float16 == txt.dtype
    assert list(meta["caption"]) == [f"caption {j}" for j in range(1000)]
    assert np.allclose(np.linalg.norm(img.astype(np.float32), axis=1), 1, atol=2e-3)
    assert json.loads((out / "stats" / "0.json").read_text())["sample_count"] == 1000
    assert torch.get_num_threads() >= 1
