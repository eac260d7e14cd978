"""Generate tests/golden/reference_mapper_*.npz by running the REFERENCE's own ClipMapper (build container only).

`clip_retrieval/clip_inference/mapper.py` is loaded by file path and executed unmodified.  Its two imports that are not
installed here are satisfied by stubs placed in sys.modules first:
  * `all_clip.load_clip` -- the wheel (all_clip>=1.3.0,<2, requirements.txt) is absent.  Its `hf_clip:` backend
    (all_clip/hf_clip.py) wraps `transformers.CLIPModel` as  encode_image(x) = model.get_image_features(x),
    encode_text(t) = model.get_text_features(t);  the stub below is that wrapper around a transformers.CLIPModel with
    the seeded random weights of oracle/clip_oracle.py:HFClipOracle (no checkpoint exists offline).
  * `sentence_transformers` -- only used for use_mclip=True, never here.
Everything after the model call -- `/= norm(dim=-1, keepdim=True)`, `.cpu().to(torch.float16).numpy()`, the returned
dict (mapper.py:49-78) -- is the reference's code.  Inputs are the seeded synthetic batches of the oracle module
(re-derivable from their seeds), outputs are committed.  Usage: python tests/golden/make_golden_mapper.py"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/clip_retrieval/clip_inference/mapper.py"

from oracle.clip_oracle import ARCHS, HFClipOracle, normalise_u8_nhwc, synth_pixels_u8, synth_tokens  # noqa: E402

CASES = [("tiny-B/32", 3, 11, 12), ("tiny-L/14", 2, 21, 22), ("tiny-H/14", 2, 31, 32)]  # (arch, batch, pixel seed, token seed)


class HFClipWrapper(torch.nn.Module):  # all_clip/hf_clip.py
    def __init__(self, inner_model):
        super().__init__()
        self.inner_model = inner_model

    def encode_image(self, image):
        out = self.inner_model.get_image_features(image)
        return out if isinstance(out, torch.Tensor) else out.pooler_output

    def encode_text(self, text):
        out = self.inner_model.get_text_features(text)
        return out if isinstance(out, torch.Tensor) else out.pooler_output


def main():
    models = {}
    stub = types.ModuleType("all_clip")
    stub.load_clip = lambda clip_model, use_jit, warmup_batch_size, clip_cache_path: (models[clip_model], None, None)
    sys.modules["all_clip"] = stub
    st = types.ModuleType("sentence_transformers")
    st.SentenceTransformer = object
    sys.modules["sentence_transformers"] = st
    spec = importlib.util.spec_from_file_location("ref_mapper", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    for name, B, ps, ts in CASES:
        arch = ARCHS[name]
        models[name] = HFClipWrapper(HFClipOracle(arch, seed=0).model)
        mapper = ref.ClipMapper(enable_image=True, enable_text=True, enable_metadata=True, use_mclip=False, clip_model=name,
                                use_jit=False, mclip_model="", warmup_batch_size=1, clip_cache_path=None)
        pix = normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=ps))
        ids = synth_tokens(B, arch.ctx_len, arch.vocab, seed=ts)
        item = {"image_tensor": torch.from_numpy(pix), "text_tokens": torch.from_numpy(ids).long(),
                "image_filename": [f"{i}.jpg" for i in range(B)], "text": [f"caption {i}" for i in range(B)],
                "metadata": ["{}"] * B}
        out = mapper(item)
        assert out["image_embs"].dtype == np.float16 and out["image_embs"].shape == (B, arch.embed_dim)
        path = os.path.join(HERE, "reference_mapper_" + name.replace("/", "-") + ".npz")
        np.savez(path, image_embs=out["image_embs"], text_embs=out["text_embs"], batch=B, pixel_seed=ps, token_seed=ts)
        print("wrote", path, out["image_embs"].shape)


if __name__ == "__main__":
    main()
