"""CPU emulation of the HIP encoder's NUMERICS (not its kernels): fp16 residual stream, LayerNorm folded into fp16 weights,
bf16 operands everywhere else, f32 accumulation / softmax / statistics (clipx_api.hip run_layers).  Used to predict,
without a GPU, what outlier magnitudes do to the parity cosine and whether the stream leaves fp16's range.
usage: python tools/emulate_fp16_stream.py tiny-L/14 [--outliers 300,-300] [--gain 30] [--batch 2]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.clip_oracle import (ARCHS, HFClipOracle, mapper_semantics, normalise_u8_nhwc, parity_report, synth_pixels_u8,  # noqa: E402
                                synth_tokens, unpack_blob)


ABLATE = set()  # --exact stream,acts,weights,fold: leave that rounding out (which rounding costs what)


def bf(x, what="acts"):
    if "fp16ref" in ABLATE:  # what an all-fp16 model (the reference's CUDA path: model.half(), f32 accumulate) would round to
        return x.to(torch.float16).float()
    return x if what in ABLATE else x.to(torch.bfloat16).float()


def h16(x, what="stream"):
    return x if what in ABLATE else x.to(torch.float16).float()


def act(x, kind):
    return x * torch.sigmoid(1.702 * x) if kind == "quick_gelu" else torch.nn.functional.gelu(x)


def blocks(x16, layers, heads, eps, kind, causal, lens=None):
    B, T, w = x16.shape
    dh = w // heads
    peak = float(x16.abs().max())
    mask = torch.full((T, T), float("-inf")).triu_(1) if causal else None
    for L in layers:
        def folded(W, b, g, beta):
            Wg = W * g[None, :]
            return h16(Wg - Wg.mean(dim=1, keepdim=True), "fold"), b + W @ beta

        rstd = 1.0 / torch.sqrt(x16.var(dim=-1, unbiased=False, keepdim=True) + eps)
        W2, c = folded(L["qkv_w"], L["qkv_b"], L["ln1_w"], L["ln1_b"])
        qkv = (h16 if "qkv_bf16" not in ABLATE else bf)(rstd * (x16 @ W2.T) + c, "qkv")  # round 4: q, k, v are IEEE fp16
        q, k, v = [t.reshape(B, T, heads, dh).transpose(1, 2) for t in qkv.split(w, dim=-1)]
        s = (q @ k.transpose(-1, -2)) * dh ** -0.5
        if mask is not None:
            s = s + mask
        p = torch.exp(s - s.amax(dim=-1, keepdim=True))
        a = bf(((h16 if "qkv_bf16" not in ABLATE else bf)(p, "p") @ v) / p.sum(dim=-1, keepdim=True), "att").transpose(1, 2).reshape(B, T, w)
        x16 = h16(x16 + (a @ bf(L["out_w"], "weights").T + L["out_b"]))
        peak = max(peak, float(x16.abs().max()))
        rstd = 1.0 / torch.sqrt(x16.var(dim=-1, unbiased=False, keepdim=True) + eps)
        W2, c = folded(L["fc1_w"], L["fc1_b"], L["ln2_w"], L["ln2_b"])
        hbuf = bf(act(rstd * (x16 @ W2.T) + c, kind), "h")
        x16 = h16(x16 + (hbuf @ bf(L["fc2_w"], "weights").T + L["fc2_b"]))
        peak = max(peak, float(x16.abs().max()))
    return x16, peak


@torch.no_grad()
def emulate_image(W, arch, pixels):
    B = pixels.shape[0]
    P, g, w = arch.patch_size, arch.image_size // arch.patch_size, arch.v_width
    patches = bf(pixels.float().reshape(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P))
    x = patches @ bf(W["conv"], "weights").T
    x = torch.cat([W["cls"].expand(B, 1, w), x], dim=1) + W["vpos"]
    x16 = h16(torch.nn.functional.layer_norm(x, (w,), W["ln_pre_w"], W["ln_pre_b"], arch.ln_eps))
    x16, peak = blocks(x16, W["vlayers"], arch.v_heads, arch.ln_eps, arch.act, False)
    y = torch.nn.functional.layer_norm(x16[:, 0], (w,), W["ln_post_w"], W["ln_post_b"], arch.ln_eps)
    return y @ bf(W["vproj"], "weights").T, peak


@torch.no_grad()
def emulate_text(W, arch, ids):
    ids = ids.long()
    w = arch.t_width
    x16 = h16(W["tok"][ids] + W["tpos"])
    x16, peak = blocks(x16, W["tlayers"], arch.t_heads, arch.ln_eps, arch.act, True)
    y = torch.nn.functional.layer_norm(x16[torch.arange(x16.shape[0]), ids.argmax(dim=-1)], (w,), W["ln_final_w"], W["ln_final_b"], arch.ln_eps)
    return y @ bf(W["tproj"], "weights").T, peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("arch")
    ap.add_argument("--outliers", default="300,-300")
    ap.add_argument("--gain", type=float, default=30.0)
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--exact", default="", help="comma list of roundings to leave out: stream, acts, weights, fold, qkv, p, att, h; qkv_bf16 = the round-3 numerics (q, k, v, P in bf16); fp16ref = every operand fp16")
    a = ap.parse_args()
    ABLATE.update(v for v in a.exact.split(",") if v)
    arch = ARCHS[a.arch]
    o = HFClipOracle(arch, seed=a.seed, threads=os.cpu_count())
    o.make_trained_like(seed=a.seed, outliers=tuple(float(v) for v in a.outliers.split(",") if v), gain=a.gain)
    W = unpack_blob(o.export_blob(), arch)
    pix = torch.from_numpy(normalise_u8_nhwc(synth_pixels_u8(a.batch, arch.image_size, seed=11)))
    ids = torch.from_numpy(synth_tokens(a.batch, arch.ctx_len, arch.vocab, seed=12))
    for name, emu, want in (("image", emulate_image(W, arch, pix), o.encode_image(pix)), ("text", emulate_text(W, arch, ids), o.encode_text(ids))):
        got, peak = emu
        rep = parity_report(mapper_semantics(got)[1], mapper_semantics(want)[1])
        print(f"{a.arch} {name}: 1-cos max {1 - rep['cos'].min():.2e}  centred min {rep['centred'].min():.5f}  nearest ok {bool((rep['nearest'] == np.arange(a.batch)).all())}"
              f"  peak |x| {peak:.0f}  closest wrong row {rep['other'].max():.4f}", flush=True)


if __name__ == "__main__":
    main()
