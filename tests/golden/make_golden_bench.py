"""Generate tests/golden/bench_oracle_ViT-L-14_b256.npz: the CPU oracle's embeddings of EVERY row of bench.py's default batch.

bench.py's live parity pass sends `--parity-rows` (64) of the 256 timed rows through the fp32 oracle on the GPU box's host cores
(0.37 s per pair; all 256 would triple the run).  ADVICE r3: that leaves 192 rows with a finite / unit-norm check only.  This
script runs the SAME oracle (oracle/clip_oracle.py:HFClipOracle = transformers.CLIPModel fp32, the reference's hf_clip
backend, + mapper.py:58-59's normalise / fp16) once, here, over all 256 image + text rows of the default workload (weights =
encoder.random_blob(seed 0), images = synth_pixels_u8(seed 1), tokens = synth_tokens(seed 2): all re-derivable), and commits the
fp16 rows; bench.py then gates ALL rows against them and cross-checks the stored rows against its live oracle rows, so a stale
fixture cannot pass.  Usage: python tests/golden/make_golden_bench.py   (~3 minutes on 8 cores)"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from clip_retrieval_amd.encoder import ARCHS, random_blob  # noqa: E402  (host code only)
from clip_retrieval_amd.synth import normalise_u8_nhwc, synth_pixels_u8, synth_tokens  # noqa: E402
from oracle.clip_oracle import ARCHS as OARCHS, HFClipOracle, mapper_semantics  # noqa: E402


def main(model="ViT-L/14", B=256):
    arch = ARCHS[model]
    o = HFClipOracle(OARCHS[model], seed=0, threads=os.cpu_count())
    o.load_blob(random_blob(arch, seed=0))
    pix = normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=1))
    ids = synth_tokens(B, arch.ctx_len, arch.vocab, seed=2)
    img, txt = [], []
    t = time.time()
    for s in range(0, B, 8):
        img.append(mapper_semantics(o.encode_image(torch.from_numpy(pix[s:s + 8])))[0])
        txt.append(mapper_semantics(o.encode_text(torch.from_numpy(ids[s:s + 8])))[0])
        print(s, round(time.time() - t, 1), flush=True)
    path = os.path.join(HERE, "bench_oracle_" + model.replace("/", "-") + f"_b{B}.npz")
    np.savez_compressed(path, image_embs=np.concatenate(img), text_embs=np.concatenate(txt), blob_seed=0, pixel_seed=1, token_seed=2,
                        pixel_sha=np.frombuffer(__import__("hashlib").sha256(pix.tobytes()).digest(), dtype=np.uint8),
                        token_sha=np.frombuffer(__import__("hashlib").sha256(ids.tobytes()).digest(), dtype=np.uint8))
    print("wrote", path)


if __name__ == "__main__":
    main()
