"""Readers feeding the mapper: folder-of-files and webdataset-style tar shards ("next" row 8f-2).

Same constructor arguments and batch dicts as the reference readers
(clip_retrieval/clip_inference/reader.py:208-269): iterating yields
  {"image_tensor": f32 [B,3,S,S], "image_filename": [str], "text_tokens": int [B,77], "text": [str],
   "metadata": [str]}  (only the enabled keys)
with the reference's skip-bad-sample behaviour (reader.py:100-104,142,180,187-189).
Differences, stated: no `webdataset` / `torchvision` / DataLoader are used -- tar members are grouped by the
webdataset key rule (basename up to the first dot) with `tarfile`, decoded with PIL in `num_prepro_workers` worker
PROCESSES (what the reference's DataLoader workers are; chunks of raw samples go out, decoded arrays come back, input
order is kept; a preprocess / tokenizer that cannot be pickled falls back to a thread pool, which the GIL bounds at
~2 k samples/s) and collated into (pinned when a GPU is present) torch tensors.  `preprocess` defaults to a
restatement of CLIP's transform (bicubic resize of the shorter side to S, centre crop, RGB, /255,
mean/std); pass the real `preprocess`/`tokenizer` from the model package when they are available.
"""

import io
import tarfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

CLIP_MEAN = np.asarray((0.48145466, 0.4578275, 0.40821073), dtype=np.float32)
CLIP_STD = np.asarray((0.26862954, 0.26130258, 0.27577711), dtype=np.float32)
IMAGE_EXTS = ("png", "jpg", "jpeg", "bmp", "webp")


def clip_preprocess(image, size=224):
    """PIL image -> f32 [3,size,size] (CLIP `_transform`: Resize(bicubic) -> CenterCrop -> RGB -> ToTensor -> Normalize)."""
    from PIL import Image  # pylint: disable=import-outside-toplevel

    w, h = image.size
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    image = image.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    image = image.crop((left, top, left + size, top + size)).convert("RGB")
    # torchvision's arithmetic operation for operation -- ToTensor: x / 255, Normalize: (x - mean) / std, f32 -- so that the tensor is
    # the reference reader's bit for bit (tests/test_reader_reference_tensors.py: the reference-held test_tensors/*.pkl)
    x = np.asarray(image, dtype=np.float32) / np.float32(255.0)
    x = (x - CLIP_MEAN) / CLIP_STD
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def clip_preprocess_u8(image, size=224):
    """The geometric half of CLIP's transform only: PIL image -> uint8 [size, size, 3] (bicubic resize of the shorter side,
    centre crop, RGB).  The /255 and mean/std normalisation run on the GPU (CLIPX_PIX_U8_NHWC of include/clipx.h): a quarter
    of the bytes through the pinned buffers and PCIe, and no float arithmetic on the host.  A reader built
    with this preprocess yields `image_tensor` as uint8 [B, S, S, 3]; ClipMapper accepts both forms."""
    from PIL import Image  # pylint: disable=import-outside-toplevel

    w, h = image.size
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    image = image.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    return np.asarray(image.crop((left, top, left + size, top + size)).convert("RGB"), dtype=np.uint8)


class DecodeRgbU8:
    """No geometry on the host: PIL image -> its decoded RGB pixels, uint8 [h, w, 3] of whatever size it has.  A reader built with
    this preprocess hands batches over as `image_raw` (one packed page-locked buffer + per-image offsets and sizes) and the
    resize / centre crop run on the GPU, bit-identical to Pillow (csrc/preprocess.hip, clipx_resize_crop_u8_device; SURVEY 8 row
    f2).  Worth it when the host's decode processes are the bottleneck and the sources are not much larger than the crop: the
    decoded source travels through the pipes and PCIe instead of the 150 KB crop.

    "Bit-identical" holds for what the GPU kernel restates: the 8-bit bicubic resample of RGB (and L: three equal channels)
    images.  Pillow resamples the other modes DIFFERENTLY from their RGB conversion -- palette (P) and bilevel (1) images with
    NEAREST, RGBA / LA / PA with premultiplied alpha, I / F / I;16 in 32-bit arithmetic -- and the reference (like
    clip_preprocess_u8) resizes in the image's own mode and converts afterwards (CLIP's `_transform`: Resize, CenterCrop,
    `_convert_image_to_rgb`).  Those images, and sources beyond what the kernel takes (a down-scale above ~30 x or a row that does
    not fit its LDS staging: CLIPX_E_UNSUPPORTED would otherwise abort the whole partition), go through the host transform here
    and travel as a ready size x size crop, which the GPU resample passes through unchanged."""

    raw_images = True  # readers: collate as `image_raw`, not as a stacked `image_tensor`
    GPU_MODES = ("RGB", "L")

    def __init__(self, size=224, max_side=8192, max_downscale=16):
        self.size, self.max_side, self.max_downscale = size, max_side, max_downscale

    def __call__(self, image):
        w, h = image.size
        if image.mode not in self.GPU_MODES or max(w, h) > self.max_side or min(w, h) > self.max_downscale * self.size:
            return clip_preprocess_u8(image, self.size)
        return np.asarray(image.convert("RGB"), dtype=np.uint8)


decode_rgb_u8 = DecodeRgbU8(224)  # the 224-pixel models' instance (worker() builds one for the model's own image size)


class ClipTransform:
    """`preprocess` as `load_clip` returns it (all_clip's second result; the reference uses it at reader.py:83,144 and
    clip_back.py:241): PIL image -> torch f32 [3, size, size].  A class, not a closure, so that it can be pickled."""

    def __init__(self, size=224):
        self.size = size

    def __call__(self, image):
        import torch  # pylint: disable=import-outside-toplevel

        return torch.from_numpy(clip_preprocess(image, self.size))


class HashTokenizer:
    """Deterministic stand-in for CLIP's BPE tokenizer, FOR SYNTHETIC DATA ONLY (the BPE merges file ships
    inside the `clip` wheel, which is not available offline): words -> ids by FNV hash, SOT ... EOT, zero pad."""

    def __init__(self, ctx_len=77, vocab=49408):
        self.ctx_len, self.vocab = ctx_len, vocab

    def tokenize_numpy(self, texts):
        out = np.zeros((len(texts), self.ctx_len), dtype=np.int64)
        for i, t in enumerate(texts):
            ids = []
            for word in t.lower().split()[: self.ctx_len - 2]:
                h = 2166136261
                for ch in word.encode("utf-8"):
                    h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
                ids.append(1 + h % (self.vocab - 3))
            seq = [self.vocab - 2] + ids + [self.vocab - 1]
            out[i, : len(seq)] = seq
        return out

    def __call__(self, texts):
        import torch  # pylint: disable=import-outside-toplevel

        return torch.from_numpy(self.tokenize_numpy(texts))


def folder_to_keys(folder, enable_text=True, enable_image=True, enable_metadata=False):
    """Keys of a folder dataset, in the reference's order (reader.py:10-51).

    One enabled modality (the reference's tested configuration, test_reader.py): exactly the reference's rule -- a key is
    the file's path relative to the folder INCLUDING its extension, keys sorted as strings ('a-b.jpg' < 'a.png').
    Several modalities: the reference intersects full paths of different extensions, i.e. it cannot pair `x.jpg` with
    `x.txt` at all in this snapshot (KeyError at reader.py:97); here files are paired by the path without its suffix, the
    order is that of the first enabled modality's full paths (text, then image, then metadata -- the reference's
    priority), and two files of one modality that share a stem (a.jpg + a.png) are reported instead of silently dropped.
    Returns (keys, text_files, image_files, metadata_files) with the three maps keyed by `keys`."""
    root = Path(folder)

    def index(exts):
        found = {}
        for ext in exts:
            for variant in (ext, ext.upper()):
                for p in root.glob(f"**/*.{variant}"):
                    found[p.relative_to(root).as_posix()] = p
        return found

    text_files = index(("txt",)) if enable_text else None
    image_files = index(IMAGE_EXTS) if enable_image else None
    metadata_files = index(("json",)) if enable_metadata else None
    enabled = [m for m in (text_files, image_files, metadata_files) if m is not None]
    if len(enabled) <= 1:
        keys = sorted(enabled[0]) if enabled else []
        return keys, text_files, image_files, metadata_files

    def by_stem(m):
        out = {}
        for full in sorted(m):
            stem = full.rsplit(".", 1)[0]
            if stem in out:
                print(f"folder_to_keys: {full} and {out[stem][0]} share a key; keeping {out[stem][0]}")
                continue
            out[stem] = (full, m[full])
        return out

    stems = [by_stem(m) for m in enabled]
    common = set(stems[0])
    for st in stems[1:]:
        common &= set(st)
    keys = [st for st in sorted(stems[0], key=lambda k: stems[0][k][0]) if st in common]
    remap = lambda m: None if m is None else {k: v[1] for k, v in by_stem(m).items() if k in common}
    return keys, remap(text_files), remap(image_files), remap(metadata_files)


def _collate(samples, enable_image, enable_text, enable_metadata, pin):
    import torch  # pylint: disable=import-outside-toplevel

    batch = {}
    if enable_image and "image_raw" in samples[0]:
        # decoded sources of different sizes: ONE packed (page-locked) byte buffer + host-side offsets and (h, w) per image --
        # the layout clipx_resize_crop_u8_device takes
        sizes = np.asarray([s["image_raw"].shape[:2] for s in samples], dtype=np.int32)
        nbytes = sizes[:, 0].astype(np.int64) * sizes[:, 1] * 3
        offsets = np.zeros(len(samples), dtype=np.int64)
        np.cumsum(nbytes[:-1], out=offsets[1:])
        packed = torch.empty(int(nbytes.sum()), dtype=torch.uint8, pin_memory=bool(pin))
        flat = packed.numpy()
        for s, o, n in zip(samples, offsets, nbytes):
            flat[o:o + n] = s["image_raw"].reshape(-1)
        batch["image_raw"] = {"pixels": packed, "offsets": offsets, "hw": sizes}
        batch["image_filename"] = [s["image_filename"] for s in samples]
    elif enable_image:
        first = samples[0]["image_tensor"]
        if pin:  # rows go straight into the page-locked batch (one copy instead of stack + pin_memory)
            t = torch.empty((len(samples),) + first.shape, dtype=torch.from_numpy(np.empty(0, first.dtype)).dtype, pin_memory=True)
            np.stack([s["image_tensor"] for s in samples], out=t.numpy())
        else:
            t = torch.from_numpy(np.stack([s["image_tensor"] for s in samples]))
        batch["image_tensor"] = t
        batch["image_filename"] = [s["image_filename"] for s in samples]
    if enable_text:
        batch["text_tokens"] = torch.from_numpy(np.stack([s["text_tokens"] for s in samples]))
        batch["text"] = [s["text"] for s in samples]
    if enable_metadata:
        batch["metadata"] = [s["metadata"] for s in samples]
    return batch


class _Span(tuple):
    """(path, offset, size): bytes of a local file that the DECODING side reads (os.pread), so that image bytes never pass
    through the parent process or its pipes."""

    __slots__ = ()


_span_fds = {}


def _read_span(span):
    import os  # pylint: disable=import-outside-toplevel

    path, off, size = span
    if size < 0:  # a whole file (FilesReader): no descriptor kept
        with open(path, "rb") as f:
            return f.read()
    fd = _span_fds.get(path)
    if fd is None:
        if len(_span_fds) >= 64:
            for f in _span_fds.values():
                os.close(f)
            _span_fds.clear()
        fd = _span_fds[path] = os.open(path, os.O_RDONLY)
    out = os.pread(fd, size, off)
    if len(out) != size:
        raise OSError(f"{path}: short read at {off} (+{size})")
    return out


def _resolve_spans(raw):
    if isinstance(raw.get("image"), _Span):
        raw["image"] = _read_span(raw["image"])
    for k in ("text", "metadata"):
        if isinstance(raw.get(k), _Span):
            raw[k] = _read_span(raw[k]).decode("utf-8")
    return raw


def _scan_plain_tar(path):
    """(member name, data offset, size) of every regular file of an UNCOMPRESSED local tar, from the 512-byte headers alone
    (no member data is read).  Returns None for anything this quick parser does not cover -- not a ustar / GNU archive,
    GNU long names, pax extended headers, sparse files -- and the caller streams the shard through `tarfile` instead."""
    import os  # pylint: disable=import-outside-toplevel

    out = []
    try:
        fd = os.open(path, os.O_RDONLY)
    except OSError:
        return None
    try:
        end = os.fstat(fd).st_size
        off = 0
        while off + 512 <= end:
            h = os.pread(fd, 512, off)
            if len(h) < 512:
                return None
            if h == b"\0" * 512:
                break
            if h[257:262] != b"ustar":
                return None
            t = h[156:157]
            sz = h[124:136]
            if sz[0] & 0x80:  # base-256 size (members over 8 GiB)
                size = int.from_bytes(sz[1:], "big")
            else:
                size = int(sz.split(b"\0", 1)[0].strip() or b"0", 8)
            if t in (b"0", b"\0", b"7"):
                name = h[0:100].split(b"\0", 1)[0]
                prefix = h[345:500].split(b"\0", 1)[0] if h[257:263] == b"ustar\0" else b""
                out.append(((prefix + b"/" + name if prefix else name).decode("utf-8"), off + 512, size))
            elif t in (b"L", b"K", b"x", b"g", b"S"):
                return None
            else:
                size = size if t not in (b"1", b"2", b"3", b"4", b"5", b"6") else 0  # links, devices, directories carry no data
            off += 512 + ((size + 511) // 512) * 512
        return out
    except (OSError, ValueError, UnicodeDecodeError):
        return None
    finally:
        os.close(fd)


def _decode_sample(raw, preprocess, tokenizer, enable_image, enable_text, enable_metadata):
    """raw: {"key", "image": bytes|None, "text": str|None, "metadata": str|None} -> sample dict of numpy arrays / strings, or
    None for an undecodable image (reference reader.py:100-104: print and skip)."""
    from PIL import Image, UnidentifiedImageError  # pylint: disable=import-outside-toplevel

    out = {}
    try:
        raw = _resolve_spans(raw)
    except (OSError, UnicodeDecodeError) as e:
        print(f"Failed to read sample {raw['key']}. Error: {e}. Skipping.")
        return None
    if enable_image:
        try:
            img = preprocess(Image.open(io.BytesIO(raw["image"])))
        except (UnidentifiedImageError, OSError, ValueError) as e:
            print(f"Failed to load image {raw['key']}. Error: {e}. Skipping.")
            return None
        img = img.numpy() if hasattr(img, "numpy") else np.asarray(img)
        if getattr(preprocess, "raw_images", False):
            out["image_raw"] = np.ascontiguousarray(img, dtype=np.uint8)  # decoded source of its own size: resized on the GPU
        else:
            out["image_tensor"] = img if img.dtype == np.uint8 else img.astype(np.float32, copy=False)  # uint8 HWC: normalised on the GPU
        out["image_filename"] = raw["key"]
    if enable_text:
        out["text"] = raw["text"]
        tok = tokenizer.tokenize_numpy([raw["text"]])[0] if hasattr(tokenizer, "tokenize_numpy") else tokenizer([raw["text"]])[0]
        out["text_tokens"] = tok.numpy() if hasattr(tok, "numpy") else np.asarray(tok)
    if enable_metadata:
        out["metadata"] = raw["metadata"]
    return out


class _DecodeWorker:
    """One `_decode_worker` child and its two pipes."""

    ARENA_BYTES = 48 << 20  # decoded pixels of one chunk come back through this shared mapping, not through the pipe

    def __init__(self, blob):
        import mmap  # pylint: disable=import-outside-toplevel
        import os  # pylint: disable=import-outside-toplevel
        import subprocess  # pylint: disable=import-outside-toplevel
        import sys  # pylint: disable=import-outside-toplevel

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # where the `clip_retrieval_amd` import shim lives
        env = dict(os.environ)
        env["PYTHONPATH"] = os.pathsep.join([root] + [p for p in sys.path if p] + [env.get("PYTHONPATH", "")])
        # an anonymous memory file shared with the child (memfd: not limited by the size of /dev/shm in a container); the child
        # writes the decoded arrays into it and the reply carries (offset, shape) instead of 150 KB of pickled pixels per image
        self.arena, fds = None, ()
        try:
            fd = os.memfd_create("clipx-decode-arena")
            os.ftruncate(fd, self.ARENA_BYTES)
            self.arena = mmap.mmap(fd, self.ARENA_BYTES)
            self._arena_fd = fd
            env["CLIPX_DECODE_ARENA"] = f"{fd}:{self.ARENA_BYTES}"
            fds = (fd,)
        except (AttributeError, OSError):
            env.pop("CLIPX_DECODE_ARENA", None)
        self.proc = subprocess.Popen([sys.executable, "-m", "clip_retrieval_amd._decode_worker"], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, env=env, pass_fds=fds)
        self._send_blob(blob)

    def _send_blob(self, blob):
        import struct  # pylint: disable=import-outside-toplevel

        self.proc.stdin.write(struct.pack("<Q", len(blob)))
        self.proc.stdin.write(blob)
        self.proc.stdin.flush()

    def recv(self):
        import pickle  # pylint: disable=import-outside-toplevel
        import struct  # pylint: disable=import-outside-toplevel

        hdr = self.proc.stdout.read(8)
        if len(hdr) < 8:
            raise RuntimeError("decode worker died")
        status, value = pickle.loads(self.proc.stdout.read(struct.unpack("<Q", hdr)[0]))
        if status != "ok":
            raise RuntimeError(f"decode worker: {value}")
        return value

    def roundtrip(self, raws):
        import pickle  # pylint: disable=import-outside-toplevel

        self._send_blob(pickle.dumps(raws, protocol=pickle.HIGHEST_PROTOCOL))
        out = self.recv()
        if self.arena is not None:
            for sample in out:  # copy the pixels out before this worker's next chunk overwrites the arena
                if sample is None:
                    continue
                for k in ("image_tensor", "image_raw"):
                    v = sample.get(k)
                    if isinstance(v, tuple) and v and v[0] == "@arena":
                        _, off, shape, dtype = v
                        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
                        sample[k] = np.frombuffer(self.arena, dtype=dtype, count=n // np.dtype(dtype).itemsize, offset=off).reshape(shape).copy()
        return out

    def close(self):
        import os  # pylint: disable=import-outside-toplevel

        try:
            self.proc.stdin.close()
            self.proc.wait(timeout=5)
        except Exception:  # pylint: disable=broad-except
            self.proc.kill()
        if self.arena is not None:
            try:
                self.arena.close()
                os.close(self._arena_fd)
            except (OSError, ValueError, BufferError):
                pass
            self.arena = None


class _DecodePool:
    """`num_prepro_workers` decode processes, started once per (preprocess, tokenizer, flags) and reused by every reader of
    this process (Runner builds a new reader per partition, runner.py:30).  A thread per worker moves the chunks through the
    pipes (pipe reads and writes release the GIL)."""

    _pools = {}

    def __init__(self, workers, blob):
        import queue  # pylint: disable=import-outside-toplevel

        self.workers = [_DecodeWorker(blob) for _ in range(workers)]
        for w in self.workers:
            w.recv()  # the worker unpickled the transform and the tokenizer
        self.idle = queue.Queue()
        for w in self.workers:
            self.idle.put(w)
        self.threads = ThreadPoolExecutor(workers)

    def _run(self, raws):
        w = self.idle.get()
        try:
            return w.roundtrip(raws)
        finally:
            self.idle.put(w)

    def submit(self, raws):
        return self.threads.submit(self._run, raws)

    def close(self):
        self.threads.shutdown(wait=False)
        for w in self.workers:
            w.close()

    @classmethod
    def get(cls, workers, args):
        """The pool for these decode arguments, or None when they cannot travel to another process (a lambda, a transform
        defined in the main script): the caller then decodes on threads."""
        import atexit  # pylint: disable=import-outside-toplevel
        import pickle  # pylint: disable=import-outside-toplevel

        try:
            blob = pickle.dumps(args, protocol=pickle.HIGHEST_PROTOCOL)
        except Exception:  # pylint: disable=broad-except
            return None
        key = (workers, blob)
        if key not in cls._pools:
            if not cls._pools:
                atexit.register(cls.shutdown)
            try:
                cls._pools[key] = cls(workers, blob)
            except (RuntimeError, OSError) as e:
                print(f"reader: decode processes unavailable ({e}); decoding on threads")
                cls._pools[key] = None
        return cls._pools[key]

    @classmethod
    def shutdown(cls):
        for pool in cls._pools.values():
            if pool is not None:
                pool.close()
        cls._pools.clear()


class _DecodedStream:
    """Iterator over a reader's decoded samples (see _BatchingReader._decoded)."""

    def __init__(self, reader):
        from collections import deque  # pylint: disable=import-outside-toplevel

        r = self.r = reader
        self.procs = _DecodePool.get(r.workers, r._decode_args()) if r.use_processes and r.workers > 1 else None  # pylint: disable=protected-access
        self.raws = iter(r._raw_samples())  # pylint: disable=protected-access
        self.inflight, self.ready, self.done = deque(), deque(), False
        if self.procs is not None:
            self.limit = r.inflight_chunks or (2 * r.workers + max(1, r.batch_size // r.chunk))  # chunks in flight
            self.pool = None
        else:
            self.limit = max(2 * r.workers, 2) + r.batch_size  # samples in flight
            self.pool = ThreadPoolExecutor(r.workers)

    def prime(self, cap=None):
        """Submit work until the window (or `cap` entries of it) is full; never blocks on a result."""
        r = self.r
        limit = self.limit if cap is None else min(self.limit, cap)
        while not self.done and len(self.inflight) < limit:
            if self.procs is not None:
                part = []
                for raw in self.raws:
                    part.append(raw)
                    if len(part) == r.chunk:
                        break
                if len(part) < r.chunk:
                    self.done = True
                if part:
                    self.inflight.append(self.procs.submit(part))
            else:
                raw = next(self.raws, None)
                if raw is None:
                    self.done = True
                    break
                self.inflight.append(self.pool.submit(r._decode, raw))  # pylint: disable=protected-access

    def __iter__(self):
        return self

    def __next__(self):
        while not self.ready:
            self.prime()
            if not self.inflight:
                if self.pool is not None:
                    self.pool.shutdown(wait=False)
                    self.pool = None
                raise StopIteration
            res = self.inflight.popleft().result()
            if self.procs is not None:
                self.ready.extend(res)
            else:
                self.ready.append(res)
            self.prime()  # refill the window behind the chunk that just left it, before the caller goes away with the sample
        return self.ready.popleft()


class _BatchingReader:
    def __init__(self, preprocess, tokenizer, batch_size, num_prepro_workers, enable_text, enable_image, enable_metadata):
        self.preprocess = preprocess or clip_preprocess
        self.tokenizer = tokenizer
        self.batch_size = batch_size
        self.workers = max(1, num_prepro_workers)
        self.num_streams = num_prepro_workers
        self.enable_text, self.enable_image, self.enable_metadata = enable_text, enable_image, enable_metadata
        self.use_processes = True  # False: decode on a thread pool inside this process
        if enable_text and tokenizer is None:
            raise ValueError("enable_text needs a tokenizer (pass the model package's, or HashTokenizer for synthetic data)")
        try:
            import torch  # pylint: disable=import-outside-toplevel

            self.pin = torch.cuda.is_available()
        except ImportError:
            self.pin = False

    # samples per task sent to a decode process: amortises the per-task pickling / wake-up cost (a task per sample caps the
    # parent at a few thousand samples/s)
    chunk = 16
    inflight_chunks = 0  # 0: 2 x workers + a batch's worth; WebdatasetReader's per-stream sub-readers set two batches' worth

    def _decode_args(self):
        return (self.preprocess, self.tokenizer, self.enable_image, self.enable_text, self.enable_metadata)

    def _decode(self, raw):
        return _decode_sample(raw, *self._decode_args())

    def _raw_samples(self):
        raise NotImplementedError

    # True: group `batch_size` RAW samples first and drop the failed ones from each group -- what the reference's
    # FilesReader does (DataLoader batches dataset indices, its collate_fn then filters the `None`s: reader.py:187-189),
    # so a failed image makes THAT batch short.  False: drop failed samples first, then batch -- the reference's
    # webdataset pipeline (map(..., handler=warn_and_continue) before batching: reader.py:142-180).
    batch_before_filter = False

    def _decoded(self):
        """Decoded samples in input order with a BOUNDED number of samples in flight (the reference's DataLoader holds at
        most prefetch_factor * workers batches): raw bytes are only read when a slot is free, so a 10 k-member shard is
        never resident at once (`Executor.map` would submit -- i.e. read and decode -- the whole partition up front).
        Returns an iterator with a `prime()` method: submit the first window of work WITHOUT waiting for any of it (the
        multi-stream WebdatasetReader primes all its streams before it pulls the first batch)."""
        return _DecodedStream(self)

    def _batches(self, stream):
        pending, taken = [], 0
        for decoded in stream:
            taken += 1
            if decoded is not None:
                pending.append(decoded)
            full = taken == self.batch_size if self.batch_before_filter else len(pending) == self.batch_size
            if full:
                if pending:
                    yield _collate(pending, self.enable_image, self.enable_text, self.enable_metadata, self.pin)
                pending, taken = [], 0
        if pending:
            yield _collate(pending, self.enable_image, self.enable_text, self.enable_metadata, self.pin)

    def __iter__(self):
        return self._batches(self._decoded())


class FilesReader(_BatchingReader):
    """Reads image / .txt / .json files sharing a stem from a folder tree."""

    batch_before_filter = True

    def __init__(self, sampler, preprocess, tokenizer, input_dataset, batch_size, num_prepro_workers,
                 enable_text=True, enable_image=True, enable_metadata=False):
        super().__init__(preprocess, tokenizer, batch_size, num_prepro_workers, enable_text, enable_image, enable_metadata)
        keys, self.text_files, self.image_files, self.metadata_files = folder_to_keys(
            input_dataset, enable_text, enable_image, enable_metadata)
        self.keys = sampler(keys)

    def _raw_samples(self):
        for k in self.keys:
            raw = {"key": k, "image": None, "text": None, "metadata": None}
            if self.enable_image:
                raw["key"] = str(self.image_files[k])
                raw["image"] = _Span((str(self.image_files[k]), 0, -1))  # read by the decoding side, not here
            if self.enable_text:
                raw["text"] = self.text_files[k].read_text()
            if self.enable_metadata:
                raw["metadata"] = self.metadata_files[k].read_text()
            yield raw


class WebdatasetReader(_BatchingReader):
    """Reads webdataset-format tar shards: consecutive members sharing `<key>.` form one sample."""

    def __init__(self, sampler, preprocess, tokenizer, input_dataset, batch_size, num_prepro_workers,
                 enable_text=True, enable_image=True, enable_metadata=False, wds_image_key="jpg",
                 wds_caption_key="txt", cache_path=None):
        super().__init__(preprocess, tokenizer, batch_size, num_prepro_workers, enable_text, enable_image, enable_metadata)
        self.cache_path = cache_path  # webdataset's cache_dir (reader.py:138): remote shards are copied here once
        shards = [input_dataset] if isinstance(input_dataset, str) else list(input_dataset)
        self.shards = sampler(shards)
        self.image_key, self.caption_key = wds_image_key, wds_caption_key

    # The reference's batch ORDER (reader.py:184-205, 262-269): its webdataset pipeline is an IterableDataset behind
    # DataLoader(num_workers = num_prepro_workers, batch_size): webdataset deals the shards to the loader workers (shard i ->
    # worker i mod W), every worker batches ITS OWN stream (a batch never mixes two workers' shards; each worker ends with its own
    # short batch), and the loader hands the batches out round-robin over the workers, skipping the ones that are exhausted.
    # Reproduced here so that the rows of img_emb_*.npy / metadata_*.parquet come out in the reference's order for the same
    # arguments -- pinned against the reference-held test_tensors/{0..3}.pkl (tests/test_reader_reference_tensors.py).  The W
    # streams share this reader's pool of decode processes.  False (or num_prepro_workers <= 1): one stream, shards in order.
    reference_batch_order = True

    def __iter__(self):
        W = self.num_streams
        if not self.reference_batch_order or W <= 1:
            yield from super().__iter__()
            return
        import copy  # pylint: disable=import-outside-toplevel

        streams, decoded = [], []
        for w in range(W):
            sub = copy.copy(self)
            sub.shards = self.shards[w::W]
            sub.reference_batch_order = False
            # one batch's worth (+ 1 chunk) in flight per stream: it is submitted when the stream's previous batch is handed out and
            # has a whole round of the other streams' batches to get decoded.  (Two batches' worth -- the reference's
            # prefetch_factor -- kept 4 k decoded samples alive across 8 streams: the parent's copies ran out of cache, -3 .. -7 %)
            sub.inflight_chunks = max(2, (self.batch_size + self.chunk - 1) // self.chunk + 1)
            if sub.shards:
                d = sub._decoded()
                decoded.append(d)
                streams.append(sub._batches(d))
        # every stream's first window goes to the decode processes before the first batch is waited for -- one batch's worth per
        # stream first (the order the batches will be asked for), then the rest: without this the W first batches were decoded one
        # after the other (-14 % on a 16 k-sample partition, profiles/r04e_pipeline_vitl14*.log)
        first = (self.batch_size + self.chunk - 1) // self.chunk
        for d in decoded:
            d.prime(first)
        for d in decoded:
            d.prime()
        while streams:
            for it in list(streams):
                batch = next(it, None)
                if batch is None:
                    streams.remove(it)
                else:
                    yield batch

    def _emit(self, key, fields):
        if self.enable_image and self.image_key not in fields:
            return None
        if self.enable_text and self.caption_key not in fields:
            return None
        if self.enable_metadata and "json" not in fields:
            return None
        return {"key": key, "image": fields.get(self.image_key),
                "text": fields[self.caption_key].decode("utf-8") if self.enable_text else None,
                "metadata": fields["json"].decode("utf-8") if self.enable_metadata else None}

    def _local_copy(self, shard):
        """The local file a shard can be read from: the path itself, or the copy of a URL (s3://, gs://, https://, ...) in
        `cache_path`, made once through fsspec (the reference passes cache_dir to webdataset, reader.py:138); None for a URL
        without a cache (streamed)."""
        if "://" not in shard:
            return shard
        if not self.cache_path:
            return None
        import hashlib  # pylint: disable=import-outside-toplevel
        import os  # pylint: disable=import-outside-toplevel

        import fsspec  # pylint: disable=import-outside-toplevel

        os.makedirs(self.cache_path, exist_ok=True)
        local = os.path.join(self.cache_path, hashlib.sha1(shard.encode()).hexdigest()[:16] + "_" + shard.rsplit("/", 1)[-1])
        if not os.path.exists(local):
            with fsspec.open(shard, "rb") as src, open(local + ".part", "wb") as dst:
                while True:
                    chunk = src.read(1 << 22)
                    if not chunk:
                        break
                    dst.write(chunk)
            os.replace(local + ".part", local)
        return local

    def _open_shard(self, shard, local=None):
        """Streaming `tarfile` over the local file (path or cached copy), or over the fsspec stream of an uncached URL."""
        if local is not None:
            return tarfile.open(local, "r|*")
        import fsspec  # pylint: disable=import-outside-toplevel

        return tarfile.open(fileobj=fsspec.open(shard, "rb").open(), mode="r|*")

    # local uncompressed shards: member offsets from the headers, bytes read by the decoding side (False: always `tarfile`)
    scan_spans = True

    def _span_samples(self, shard, members):
        """The grouping of `_raw_samples` over (name, offset, size) triples; image / text / metadata stay `_Span`s."""
        cur_key, fields = None, {}

        def emit():
            if self.enable_image and self.image_key not in fields:
                return None
            if self.enable_text and self.caption_key not in fields:
                return None
            if self.enable_metadata and "json" not in fields:
                return None
            return {"key": cur_key, "image": fields.get(self.image_key),
                    "text": fields[self.caption_key] if self.enable_text else None,
                    "metadata": fields["json"] if self.enable_metadata else None}

        for name, off, size in members:
            base = name.rsplit("/", 1)
            stem, _, ext = base[-1].partition(".")
            key = (base[0] + "/" if len(base) == 2 else "") + stem
            if key != cur_key:
                if cur_key is not None:
                    s = emit()
                    if s is not None:
                        yield s
                cur_key, fields = key, {}
            fields[ext.lower()] = _Span((shard, off, size))
        if cur_key is not None:
            s = emit()
            if s is not None:
                yield s

    def _raw_samples(self):
        for shard in self.shards:
            try:
                local = self._local_copy(shard)
                if self.scan_spans and local is not None:
                    members = _scan_plain_tar(local)
                    if members is not None:
                        yield from self._span_samples(local, members)
                        continue
                tf = self._open_shard(shard, local)
            except (tarfile.TarError, OSError) as e:
                print(f"warn_and_continue: {shard}: {e}")
                continue
            with tf:
                cur_key, fields = None, {}
                try:
                    for member in tf:
                        if not member.isfile():
                            continue
                        base = member.name.rsplit("/", 1)
                        stem, _, ext = base[-1].partition(".")
                        key = (base[0] + "/" if len(base) == 2 else "") + stem
                        if key != cur_key:
                            if cur_key is not None:
                                s = self._emit(cur_key, fields)
                                if s is not None:
                                    yield s
                            cur_key, fields = key, {}
                        fields[ext.lower()] = tf.extractfile(member).read()
                except (tarfile.TarError, OSError) as e:
                    print(f"warn_and_continue: {shard}: {e}")
                if cur_key is not None:
                    s = self._emit(cur_key, fields)
                    if s is not None:
                        yield s
