"""CPU oracle for the GPU image preprocess.  TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module.  It restates, in numpy integer arithmetic, what the reference's image transform does to
the pixels before the encoder sees them (reader.py:83,87 `self.image_transform(image)` = CLIP's `_transform`, third party:
torchvision `Resize(n_px, interpolation=BICUBIC)` -> `CenterCrop(n_px)` on a PIL image, i.e. Pillow's `Image.resize`):

  * Pillow `ImagingResample` (src/libImaging/Resample.c, Pillow 9 - 11), 8 bits per channel, bicubic (a = -0.5, support 2):
    per output coordinate a window of source pixels [xmin, xmin + xmax) with double-precision weights w((x + xmin - center + 0.5)
    / filterscale), normalised to sum 1 and converted to fixed point with 22 fractional bits (round half away from zero);
    the horizontal pass produces a uint8 image ((sum + 2^21) >> 22, clipped to 0 .. 255), the vertical pass runs over that.
    A pass whose size does not change is skipped.
  * torchvision geometry: shorter side -> S, the long side int(S * long / short); centre crop offsets int(round((dim - S) / 2)).

PINNED: tests/test_resample_oracle.py requires bit equality with Pillow itself (`clip_retrieval_amd.reader.clip_preprocess_u8`,
which calls PIL) on the reference's own test images and on synthetic sizes; the HIP kernel is then held to this restatement
and to Pillow on the GPU box.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def precompute_coeffs(in_size, out_size):
    """Resample.c: precompute_coeffs + normalize_coeffs_8bpc for the full box.  Returns (ksize, bounds [out, 2], kk int32
    [out, ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax, dtype=np.float64)
        w = _bicubic((x + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:  # the C loop's summation order
            ww += v
        if ww != 0.0:
            w = w / ww
        k = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))
        kk[xx, :xmax] = np.trunc(k).astype(np.int64).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic_u8(img, out_w, out_h):
    """Pillow Image.resize((out_w, out_h), BICUBIC) of a uint8 [h, w, c] array."""
    img = np.asarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    cur = img
    if out_w != w:
        _, bounds, kk = precompute_coeffs(w, out_w)
        out = np.empty((h, out_w, img.shape[2]), dtype=np.uint8)
        for xx in range(out_w):
            xmin, xmax = bounds[xx]
            acc = (cur[:, xmin:xmin + xmax, :].astype(np.int64) * kk[xx, :xmax].astype(np.int64)[None, :, None]).sum(axis=1)
            out[:, xx, :] = _clip8(acc + (1 << (PRECISION_BITS - 1)))
        cur = out
    if out_h != h:
        _, bounds, kk = precompute_coeffs(h, out_h)
        out = np.empty((out_h, cur.shape[1], img.shape[2]), dtype=np.uint8)
        for yy in range(out_h):
            ymin, ymax = bounds[yy]
            acc = (cur[ymin:ymin + ymax, :, :].astype(np.int64) * kk[yy, :ymax].astype(np.int64)[:, None, None]).sum(axis=0)
            out[yy] = _clip8(acc + (1 << (PRECISION_BITS - 1)))
        cur = out
    return cur


def clip_geometry(h, w, size):
    """(new_w, new_h, left, top) of CLIP's Resize(size) + CenterCrop(size) (torchvision semantics)."""
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    return nw, nh, int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))


def clip_resize_crop_u8(img, size=224):
    """uint8 [h, w, 3] RGB -> uint8 [size, size, 3]: what reader.clip_preprocess_u8 (Pillow) produces."""
    h, w, _ = img.shape
    nw, nh, left, top = clip_geometry(h, w, size)
    r = resize_bicubic_u8(img, nw, nh)
    return np.ascontiguousarray(r[top:top + size, left:left + size])
