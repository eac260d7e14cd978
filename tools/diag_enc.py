import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from clip_retrieval_amd.encoder import ARCHS, ClipEncoder, ClipArch, random_blob
from clip_retrieval_amd.synth import normalise_u8_nhwc, synth_pixels_u8, synth_tokens
name = sys.argv[1]; B = int(sys.argv[2]); layers = int(sys.argv[3])
base = ARCHS[name]
arch = ClipArch(**{**{k: getattr(base, k) for k in ClipArch.__dataclass_fields__}, "v_layers": layers, "t_layers": layers})
enc = ClipEncoder(arch, random_blob(arch, 0), 0)
pix = normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=1)); ids = synth_tokens(B, arch.ctx_len, arch.vocab, seed=2)
print("text...", flush=True); t = enc.encode_text(ids); print("text ok", np.isnan(t.astype(np.float32)).sum(), flush=True)
print("image...", flush=True); a = enc.encode_image(pix); print("image ok", np.isnan(a.astype(np.float32)).sum(), flush=True)
