"""The reader's fast path for local uncompressed tar shards (row f2 / a3): member offsets from the tar headers alone, bytes read
by the decoding side (`_Span`), decoded pixels returned through a shared memory arena instead of the pipe.  Everything it
yields must equal what the `tarfile` streaming path yields; anything the quick header parser does not cover must fall back."""
import io
import os
import tarfile

import numpy as np
import pytest
from PIL import Image

from clip_retrieval_amd import reader as R
from clip_retrieval_amd.runner import Sampler

REF_TARS = "/root/reference/tests/test_clip_inference/test_tars"


def _jpeg(seed, h=40, w=56):
    rng = np.random.default_rng(seed)
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(buf, format="JPEG", quality=90)
    return buf.getvalue()


def _write_tar(path, members, fmt=tarfile.USTAR_FORMAT, mode="w"):
    with tarfile.open(path, mode, format=fmt) as tf:
        for name, data in members:
            ti = tarfile.TarInfo(name)
            ti.size = len(data)
            tf.addfile(ti, io.BytesIO(data))


def _members(n, prefix=""):
    out = []
    for i in range(n):
        out.append((f"{prefix}{i:05d}.jpg", _jpeg(i) if i != 3 else b"not a jpeg"))  # one undecodable image
        out.append((f"{prefix}{i:05d}.txt", f"caption {i} é".encode("utf-8")))
        out.append((f"{prefix}{i:05d}.json", ('{"i": %d}' % i).encode()))
    return out


@pytest.mark.parametrize("fmt", [tarfile.USTAR_FORMAT, tarfile.GNU_FORMAT, tarfile.PAX_FORMAT])
def test_header_scan_equals_tarfile(tmp_path, fmt):
    p = str(tmp_path / "a.tar")
    members = _members(7, prefix="sub/dir/")
    _write_tar(p, members, fmt)
    got = R._scan_plain_tar(p)  # pylint: disable=protected-access
    with tarfile.open(p) as tf:
        want = [(m.name, m.offset_data, m.size) for m in tf if m.isfile()]
    assert got == want and len(got) == 21
    with open(p, "rb") as f:
        for (name, off, size), (wname, data) in zip(got, members):
            f.seek(off)
            assert name == wname and f.read(size) == data


def test_header_scan_declines_what_it_does_not_parse(tmp_path):
    members = _members(3)
    gz = str(tmp_path / "a.tar.gz")
    _write_tar(gz, members, mode="w:gz")
    assert R._scan_plain_tar(gz) is None  # pylint: disable=protected-access
    long_name = "d/" + "x" * 150 + ".jpg"  # GNU long-name / pax extended headers
    for fmt in (tarfile.GNU_FORMAT, tarfile.PAX_FORMAT):
        p = str(tmp_path / f"long{fmt}.tar")
        _write_tar(p, [(long_name, _jpeg(1)), ("d/" + "x" * 150 + ".txt", b"c")], fmt)
        assert R._scan_plain_tar(p) is None  # pylint: disable=protected-access
    assert R._scan_plain_tar(str(tmp_path / "missing.tar")) is None  # pylint: disable=protected-access
    (tmp_path / "junk.tar").write_bytes(b"\x01" * 2048)
    assert R._scan_plain_tar(str(tmp_path / "junk.tar")) is None  # pylint: disable=protected-access


def _read_all(reader):
    out = []
    for b in reader:
        out.append({k: (v.numpy().copy() if hasattr(v, "numpy") else v) for k, v in b.items()})
    return out


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x.keys() == y.keys()
        for k in x:
            if isinstance(x[k], np.ndarray):
                assert np.array_equal(x[k], y[k]), k
            else:
                assert x[k] == y[k], k


@pytest.mark.parametrize("workers,procs", [(1, False), (3, False), (3, True)])
def test_span_path_equals_the_streaming_path(tmp_path, workers, procs):
    shards = []
    for s in range(2):
        p = str(tmp_path / f"{s}.tar")
        _write_tar(p, _members(11, prefix=f"s{s}/"))
        shards.append(p)
    gz = str(tmp_path / "2.tar.gz")  # a shard the scanner declines: streamed, in the same partition
    _write_tar(gz, _members(5, prefix="z/"), mode="w:gz")
    shards.append(gz)

    def make(scan):
        r = R.WebdatasetReader(Sampler(0, 1), R.clip_preprocess_u8, R.HashTokenizer(), shards, 4, workers, enable_metadata=True)
        r.use_processes = procs
        r.scan_spans = scan
        return r

    fast, slow = _read_all(make(True)), _read_all(make(False))
    _same(fast, slow)
    assert sum(len(b["text"]) for b in fast) == 27 - 3  # one undecodable image per shard is skipped
    assert fast[0]["metadata"][0] == '{"i": 0}' and fast[0]["text"][1].endswith("é")


def test_spans_reach_the_decoder_not_the_parent(tmp_path):
    p = str(tmp_path / "a.tar")
    _write_tar(p, _members(4))
    r = R.WebdatasetReader(Sampler(0, 1), R.clip_preprocess_u8, R.HashTokenizer(), [p], 4, 1)
    raws = list(r._raw_samples())  # pylint: disable=protected-access
    assert len(raws) == 4 and all(isinstance(x["image"], R._Span) and isinstance(x["text"], R._Span) for x in raws)  # pylint: disable=protected-access
    got = R._decode_sample(raws[0], R.clip_preprocess_u8, R.HashTokenizer(), True, True, False)  # pylint: disable=protected-access
    assert got["image_tensor"].shape == (224, 224, 3) and got["text"] == "caption 0 é"


def test_decode_processes_return_pixels_through_the_arena(tmp_path):
    """Raw decoded sources (variable sizes) and fixed crops both come back through the shared arena; a chunk that does not fit
    the arena falls back to the pipe for the rest -- same arrays either way."""
    p = str(tmp_path / "a.tar")
    members = []
    for i in range(9):
        members.append((f"{i:03d}.jpg", _jpeg(i, 30 + 7 * i, 50 + 3 * i)))
        members.append((f"{i:03d}.txt", b"c"))
    _write_tar(p, members)

    def run(arena_bytes):
        R._DecodePool.shutdown()  # pylint: disable=protected-access
        old = R._DecodeWorker.ARENA_BYTES  # pylint: disable=protected-access
        R._DecodeWorker.ARENA_BYTES = arena_bytes  # pylint: disable=protected-access
        try:
            r = R.WebdatasetReader(Sampler(0, 1), R.decode_rgb_u8, R.HashTokenizer(), [p], 4, 2)
            out = _read_all(r)
        finally:
            R._DecodeWorker.ARENA_BYTES = old  # pylint: disable=protected-access
            R._DecodePool.shutdown()  # pylint: disable=protected-access
        return out

    def unpack(batches):
        imgs = []
        for b in batches:
            raw = b["image_raw"]
            flat = raw["pixels"].numpy() if hasattr(raw["pixels"], "numpy") else raw["pixels"]
            for o, (h, w) in zip(raw["offsets"], raw["hw"]):
                imgs.append(np.asarray(flat[o:o + h * w * 3]).reshape(h, w, 3).copy())
        return imgs

    big, tiny = unpack(run(48 << 20)), unpack(run(16384))  # 16 KiB holds one or two of these images per chunk
    want = [np.asarray(Image.open(io.BytesIO(members[2 * i][1])).convert("RGB")) for i in range(9)]
    assert len(big) == len(tiny) == 9
    for a, b, w in zip(big, tiny, want):
        assert np.array_equal(a, w) and np.array_equal(b, w)


@pytest.mark.skipif(not os.path.isdir(REF_TARS), reason="reference fixtures are only present in the build container")
def test_reference_tars_through_both_paths():
    tars = sorted(os.path.join(REF_TARS, f) for f in os.listdir(REF_TARS) if f.endswith(".tar"))
    for t in tars:
        with tarfile.open(t) as tf:
            want = [(m.name, m.offset_data, m.size) for m in tf if m.isfile()]
        got = R._scan_plain_tar(t)  # pylint: disable=protected-access
        assert got is None or got == want

    def make(scan):
        r = R.WebdatasetReader(Sampler(0, 1), R.clip_preprocess_u8, R.HashTokenizer(), tars, 2, 2)
        r.scan_spans = scan
        return r

    _same(_read_all(make(True)), _read_all(make(False)))


@pytest.mark.parametrize("cached", [False, True])
def test_url_shards_go_through_fsspec_and_the_cache(tmp_path, cached):
    """`input_dataset` entries with a scheme (s3://, gs://, https://; here file:// and memory://) are opened through fsspec and,
    when `cache_path` is set, copied there once (the reference hands cache_dir to webdataset, reader.py:138).  Same batches as
    the local path."""
    import fsspec

    p = str(tmp_path / "a.tar")
    _write_tar(p, _members(6))
    with open(p, "rb") as f:
        blob = f.read()
    with fsspec.open("memory://shards/b.tar", "wb") as f:
        f.write(blob)
    cache = str(tmp_path / "cache") if cached else None
    want = _read_all(R.WebdatasetReader(Sampler(0, 1), R.clip_preprocess_u8, R.HashTokenizer(), [p], 4, 1))
    for url in ("file://" + p, "memory://shards/b.tar"):
        r = R.WebdatasetReader(Sampler(0, 1), R.clip_preprocess_u8, R.HashTokenizer(), [url], 4, 1, cache_path=cache)
        _same(_read_all(r), want)
    if cached:
        files = sorted(os.listdir(cache))
        assert len(files) == 2 and all(f.endswith(".tar") and not f.endswith(".part") for f in files)
        # second pass: served from the cache even when the source is gone
        fsspec.filesystem("memory").rm("/shards/b.tar")
        r = R.WebdatasetReader(Sampler(0, 1), R.clip_preprocess_u8, R.HashTokenizer(), ["memory://shards/b.tar"], 4, 1, cache_path=cache)
        _same(_read_all(r), want)
    # a shard that cannot be opened is skipped with a warning, the rest of the partition is read (warn_and_continue)
    r = R.WebdatasetReader(Sampler(0, 1), R.clip_preprocess_u8, R.HashTokenizer(), ["memory://shards/none.tar", p], 4, 1, cache_path=cache)
    _same(_read_all(r), want)
