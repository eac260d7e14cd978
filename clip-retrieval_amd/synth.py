"""Synthetic inputs and the algorithmic-FLOP model used by bench.py.

Host-side helpers only (no arithmetic of the hot path): seeded uint8 images -> the reference's
`image_tensor` layout, CLIP-shaped token rows (SOT ... EOT 0...), and the per-sample FLOP count
(2 x MACs, no padding, no recompute) that the roofline fractions are priced with.
"""

import numpy as np

CLIP_MEAN = np.asarray((0.48145466, 0.4578275, 0.40821073), dtype=np.float32)
CLIP_STD = np.asarray((0.26862954, 0.26130258, 0.27577711), dtype=np.float32)


def synth_pixels_u8(B, size=224, seed=1):
    """Structured synthetic images (i.i.d. noise images all embed to the same point --
    inter-sample oracle cosine 0.998 -- so a cosine gate could not tell one row from another).  Sample b depends on (seed, b)
    only: a saturated background colour (a corner of the RGB cube, all eight distinct within each aligned group of eight
    samples, pulled up to 15 % towards a random colour) with a gradient, a low-frequency sinusoid, 2 - 5 filled rectangles /
    ellipses of random colours, mild per-sample noise.  Random-init CLIP towers are close to position-invariant colour
    statistics, so the dominant colour is what separates embeddings: mean inter-sample cosine ~0.72, < 0.93 within a group
    of eight (tests/test_oracle.py)."""
    out = np.empty((B, size, size, 3), dtype=np.uint8)
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    u, v = xx / size, yy / size
    for b in range(B):
        r = np.random.default_rng([int(seed), b, 0x5EED])
        corner = int(np.random.default_rng([int(seed), b // 8, 0xC0]).permutation(8)[b % 8])
        c0 = np.asarray([corner & 1, (corner >> 1) & 1, (corner >> 2) & 1], dtype=np.float64) * 255.0
        c0 = c0 + r.uniform(0, 0.15) * (r.uniform(0, 255, 3) - c0)
        c1 = c0 + r.uniform(0.1, 0.3) * (r.uniform(0, 255, 3) - c0)
        ang = r.uniform(0, 2 * np.pi)
        t = (np.cos(ang) * (u - 0.5) + np.sin(ang) * (v - 0.5)) * 1.2 + 0.5
        img = c0 + np.clip(t, 0, 1)[..., None] * (c1 - c0)
        fx, fy, ph = r.uniform(0.5, 6), r.uniform(0.5, 6), r.uniform(0, 2 * np.pi)
        img = img + (r.uniform(5, 30) * np.sin(2 * np.pi * (fx * u + fy * v) + ph))[..., None] * r.uniform(-1, 1, 3)
        for _ in range(int(r.integers(2, 6))):
            cx, cy = r.uniform(0, 1, 2)
            hw, hh = r.uniform(0.04, 0.2, 2)
            col = r.uniform(0, 255, 3)
            if r.random() < 0.5:
                m = (np.abs(u - cx) < hw) & (np.abs(v - cy) < hh)
            else:
                m = ((u - cx) / hw) ** 2 + ((v - cy) / hh) ** 2 < 1.0
            img[m] = col
        img = img + r.normal(0, r.uniform(1, 10), (size, size, 3))
        out[b] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out


def normalise_u8_nhwc(u8):
    """u8 NHWC -> f32 NCHW, /255, CLIP mean/std: the tensor the reference reader hands the mapper."""
    x = u8.astype(np.float32) / np.float32(255.0)  # torchvision: ToTensor, then Normalize -- two f32 divisions (reader.clip_preprocess)
    x = (x - CLIP_MEAN) / CLIP_STD
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def synth_tokens(B, ctx_len=77, vocab=49408, seed=2):
    rng = np.random.default_rng(seed)
    ids = np.zeros((B, ctx_len), dtype=np.int32)
    lens = rng.integers(4, ctx_len - 1, B)
    for b in range(B):
        L = int(lens[b])
        ids[b, 0] = vocab - 2                                 # SOT
        ids[b, 1:L] = rng.integers(1, vocab - 3, L - 1)
        ids[b, L] = vocab - 1                                 # EOT = max id -> argmax pooling
    return ids


def tower_gflop(arch):
    """(image GFLOP, text GFLOP) per sample: QKV + out + MLP GEMMs, QK^T and PV, patch embed, projection."""
    def tower(T, w, mlp, layers):
        return layers * (2 * T * (3 * w * w + w * w + 2 * w * mlp) + 4 * T * T * w)

    g2 = (arch.image_size // arch.patch_size) ** 2
    img = (tower(arch.v_tokens, arch.v_width, arch.v_mlp, arch.v_layers) + 2 * g2 * 3 * arch.patch_size ** 2 * arch.v_width
           + 2 * arch.v_width * arch.embed_dim)
    txt = tower(arch.ctx_len, arch.t_width, arch.t_mlp, arch.t_layers) + 2 * arch.t_width * arch.embed_dim
    return img / 1e9, txt / 1e9


def perturbed_queries(rows_f32, noise=0.1, seed=4):
    """q_j = normalise(x_j + noise * eps / sqrt(d)): queries whose planted nearest neighbour is row j."""
    rng = np.random.default_rng(seed)
    x = np.asarray(rows_f32, dtype=np.float32)
    q = x + noise * rng.standard_normal(x.shape).astype(np.float32) / np.sqrt(np.float32(x.shape[1]))
    return (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
