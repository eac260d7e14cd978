"""Diagnostic: the persistent 256x256 GEMM path (large batches) against the 128x128 kernel (CLIPX_GEMM_VARIANT=1) on a
2-layer model -- both must produce bit-identical embeddings (tests/test_clip_gpu.py only reaches the 256x256 kernel's
LayerNorm-fold / shadow-store epilogues at batch sizes the CPU oracle cannot follow)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clip_retrieval_amd.encoder import ARCHS, ClipArch, ClipEncoder, random_blob  # noqa: E402
from clip_retrieval_amd.synth import normalise_u8_nhwc, synth_pixels_u8, synth_tokens  # noqa: E402

base = ARCHS["ViT-L/14"]
arch = ClipArch(**{**{k: getattr(base, k) for k in ClipArch.__dataclass_fields__}, "v_layers": 2, "t_layers": 2})
blob = random_blob(arch, seed=0)
enc = ClipEncoder(arch, blob, 0)
os.environ["CLIPX_GEMM_VARIANT"] = "1"
ref = ClipEncoder(arch, blob, 0)
os.environ.pop("CLIPX_GEMM_VARIANT")
for B in (64, 256):
    pix = normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=1))
    ids = synth_tokens(B, arch.ctx_len, arch.vocab, seed=2)
    for name, a, b in (("image", enc.encode_image(pix), ref.encode_image(pix)), ("text", enc.encode_text(ids), ref.encode_text(ids))):
        bad = np.flatnonzero((a.view(np.uint16) != b.view(np.uint16)).any(1))
        print(f"B={B} {name}: NaN rows {int(np.isnan(a.astype(np.float32)).any(1).sum())} (128x128 path {int(np.isnan(b.astype(np.float32)).any(1).sum())}), "
              f"rows differing from the 128x128 path {bad.size} {bad[:8].tolist()}", flush=True)
