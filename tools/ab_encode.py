"""Same-box A/B of encoder build switches: two encoders in one process (the switch is read at clipx_create), timed in
alternation.  usage: python tools/ab_encode.py ENVVAR valueA valueB [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from clip_retrieval_amd.encoder import ARCHS, ClipEncoder, random_blob
from clip_retrieval_amd.synth import normalise_u8_nhwc, synth_pixels_u8, synth_tokens

var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
arch = ARCHS["ViT-L/14"]
blob = random_blob(arch, seed=0)
encs = {}
for v in (va, vb):
    os.environ[var] = v
    encs[v] = ClipEncoder(arch, blob, 0)
os.environ.pop(var)
B = 256
dev = torch.device("cuda", 0)
pix = torch.from_numpy(normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=1))).to(dev)
ids = torch.from_numpy(synth_tokens(B, arch.ctx_len, arch.vocab, seed=2)).to(dev)
outs = {v: (torch.empty(B, arch.embed_dim, dtype=torch.float16, device=dev), torch.empty(B, arch.embed_dim, dtype=torch.float16, device=dev)) for v in encs}
st = torch.cuda.current_stream(dev).cuda_stream


def step(v):
    encs[v].encode_image_device(pix.data_ptr(), B, 0, outs[v][0].data_ptr(), None, st)
    encs[v].encode_text_device(ids.data_ptr(), B, outs[v][1].data_ptr(), None, st)


for v in encs:
    for _ in range(3):
        step(v)
torch.cuda.synchronize()
res = {v: [] for v in encs}
for r in range(rounds):
    for v in encs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step(v)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 10 * 1e3)
for v in encs:
    print(f"{var}={v}: ms per step {['%.2f' % x for x in res[v]]}  median {np.median(res[v]):.2f}  -> {B / np.median(res[v]) * 1e3:.0f} samples/s", flush=True)
a, b = outs[va], outs[vb]
print("outputs identical:", bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])),
      " max |diff| image", float((a[0].float() - b[0].float()).abs().max()), "text", float((a[1].float() - b[1].float()).abs().max()))
