"""`ClipMapper` on the MI355X encoder -- the drop-in for the encode seam.

Same constructor and `__call__(item) -> dict` contract as the reference class
(clip_retrieval/clip_inference/mapper.py:16-78), which `worker.py:88-99` builds through `mapper_builder`
and `Runner.__call__` invokes once per batch (runner.py:31,44):
  in : item["image_tensor"] torch f32 [B,3,S,S] (CPU), item["text_tokens"] torch int [B,77],
       item["image_filename"] / ["text"] / ["metadata"] lists (passed through)
  out: {"image_embs": f16 [B,E] | None, "text_embs": f16 [B,E] | None, "image_filename", "text", "metadata"}
Embeddings are unit-norm rows, normalised in fp32 BEFORE the fp16 cast (mapper.py:58-59), returned as
fresh C-contiguous numpy arrays the writer may keep (writer.py:50-54).  Errors propagate (no try/except in
the reference loop, runner.py:35-62).  The model forward, the normalise and the cast all run in
lib/libclipx.so; nothing falls back to torch.
"""

import numpy as np

from .encoder import get_encoder


def normalized(a, axis=-1, order=2):
    """Row-normalise with a zero-norm guard (the helper of mapper.py:8-13 / clip_back.py:194-197)."""
    norms = np.atleast_1d(np.linalg.norm(a, order, axis))
    norms[norms == 0] = 1
    return a / np.expand_dims(norms, axis)


class ClipMapper:
    """transforms images and texts into clip embeddings (MI355X)"""

    def __init__(
        self,
        enable_image,
        enable_text,
        enable_metadata,
        use_mclip,
        clip_model,
        use_jit,
        mclip_model,
        warmup_batch_size=1,
        clip_cache_path=None,
        device=0,
    ):
        del use_jit, mclip_model  # no TorchScript on this path; mclip is refused below
        if use_mclip:
            raise NotImplementedError("use_mclip needs sentence-transformers' multilingual text tower; "
                                      "only the CLIP towers are accelerated (SURVEY 2.2: out of scope)")
        self.enable_image = enable_image
        self.enable_text = enable_text
        self.enable_metadata = enable_metadata
        self.use_mclip = use_mclip
        self.device = f"cuda:{device}"
        self._enc = get_encoder(clip_model, clip_cache_path, device)  # cached per (model, device)
        if warmup_batch_size and not getattr(self._enc, "_warm", False):
            arch = self._enc.arch
            if enable_image:
                self._enc.encode_image(np.zeros((warmup_batch_size, 3, arch.image_size, arch.image_size), np.float32))
            if enable_text:
                ids = np.zeros((warmup_batch_size, arch.ctx_len), np.int32)
                ids[:, 0], ids[:, 1] = arch.vocab - 2, arch.vocab - 1
                self._enc.encode_text(ids)
            self._enc._warm = True  # pylint: disable=protected-access

    def _encode_images(self, item):
        """`image_tensor` (the reference's batch: f32 NCHW, or uint8 NHWC crops) or `image_raw` (decoded sources of any size:
        resize + centre crop on the GPU too, reader.decode_rgb_u8)."""
        if "image_raw" in item:
            return self._enc.encode_image_raw(item["image_raw"])
        return self._enc.encode_image(item["image_tensor"])

    def submit(self, item):
        """Stage the batch and enqueue its upload + kernels (clipx_encode_*_async); returns a handle for collect().
        A caller that submits batch n+1 before collecting batch n overlaps n+1's upload with n's kernels (runner.Runner).
        Batches larger than the library's max batch fall back to the synchronous call inside collect()."""
        h = {"item": item, "img": None, "txt": None}
        try:
            if self.enable_image and "image_raw" not in item and len(item["image_tensor"]) <= self._enc.max_batch:
                h["img"] = self._enc.submit_image(item["image_tensor"])  # (decoded sources: resized + encoded in collect())
            if self.enable_text and len(item["text_tokens"]) <= self._enc.max_batch:
                h["txt"] = self._enc.submit_text(item["text_tokens"])
        except BaseException:
            self.discard(h)  # a half-submitted batch must not keep a staging slot of the (cached, long-lived) encoder
            raise
        return h

    def discard(self, h):
        """Wait for and drop whatever tickets of `h` are still outstanding (error paths: a ticket that is never waited for
        keeps one of the encoder's four staging slots busy for the life of the process)."""
        for k in ("img", "txt"):
            t = h.get(k)
            if t is not None and t.get("ticket") is not None:
                try:
                    self._enc.collect(t)
                except Exception:  # pylint: disable=broad-except
                    pass

    def collect(self, h):
        item = h["item"]
        image_embs = text_embs = image_filename = text = metadata = None
        if self.enable_image:
            image_embs = self._enc.collect(h["img"]) if h["img"] is not None else self._encode_images(item)
            image_filename = item["image_filename"]
        if self.enable_text:
            text_embs = self._enc.collect(h["txt"]) if h["txt"] is not None else self._enc.encode_text(item["text_tokens"])
            text = item["text"]
        if self.enable_metadata:
            metadata = item["metadata"]
        return {
            "image_embs": image_embs,
            "text_embs": text_embs,
            "image_filename": image_filename,
            "text": text,
            "metadata": metadata,
        }

    def __call__(self, item):
        image_embs = text_embs = image_filename = text = metadata = None
        if self.enable_image:
            image_embs = self._encode_images(item)
            image_filename = item["image_filename"]
        if self.enable_text:
            text_embs = self._enc.encode_text(item["text_tokens"])
            text = item["text"]
        if self.enable_metadata:
            metadata = item["metadata"]
        return {
            "image_embs": image_embs,
            "text_embs": text_embs,
            "image_filename": image_filename,
            "text": text,
            "metadata": metadata,
        }
