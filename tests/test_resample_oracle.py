"""The resize / centre-crop oracle (oracle/resample_oracle.py) against Pillow itself -- the third-party code the reference's image
transform runs (reader.py:83,87 -> CLIP `_transform` -> torchvision Resize(BICUBIC) + CenterCrop on PIL images).  Bit equality:
this pins the restatement the HIP kernel (csrc/preprocess.hip) is then held to."""
import os

import numpy as np
import pytest
from PIL import Image

from clip_retrieval_amd import reader
from oracle import resample_oracle as ro

REF_IMAGES = "/root/reference/tests/test_clip_inference/test_images"


def _synthetic(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (max(2, h // 7), max(2, w // 7), 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR), dtype=np.int16)
    img = img + rng.integers(-40, 41, img.shape)  # smooth structure + noise: exercises both the clip8 ends and the lobes
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w", [(224, 224), (256, 256), (300, 451), (451, 300), (97, 131), (131, 97), (224, 500), (500, 224),
                                 (32, 32), (17, 400), (1024, 768), (225, 223)])
@pytest.mark.parametrize("size", [224, 64])
def test_oracle_equals_pillow_synthetic(h, w, size):
    img = _synthetic(h, w, h * 1000 + w)
    want = reader.clip_preprocess_u8(Image.fromarray(img), size=size)
    got = ro.clip_resize_crop_u8(img, size)
    assert got.shape == (size, size, 3) and got.dtype == np.uint8
    assert np.array_equal(got, np.asarray(want))


def test_oracle_equals_pillow_resize_only():
    """Image.resize itself (both passes, up and down), without the crop."""
    img = _synthetic(120, 200, 5)
    for ow, oh in ((224, 134), (50, 30), (200, 300), (400, 120), (37, 211)):
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(ro.resize_bicubic_u8(img, ow, oh), want), (ow, oh)


def test_coefficients_are_normalised_fixed_point():
    for n_in, n_out in ((640, 224), (100, 224), (224, 224), (1000, 37)):
        ksize, bounds, kk = ro.precompute_coeffs(n_in, n_out)
        assert kk.shape == (n_out, ksize)
        s = kk.astype(np.int64).sum(axis=1)
        assert np.all(np.abs(s - (1 << ro.PRECISION_BITS)) <= ksize)  # each row sums to 1.0 up to one rounding per tap
        assert np.all(bounds[:, 0] >= 0) and np.all(bounds[:, 0] + bounds[:, 1] <= n_in)


def test_geometry_is_torchvision():
    assert ro.clip_geometry(480, 640, 224) == (298, 224, 37, 0)
    assert ro.clip_geometry(640, 480, 224) == (224, 298, 0, 37)
    assert ro.clip_geometry(224, 224, 224) == (224, 224, 0, 0)
    assert ro.clip_geometry(100, 333, 224) == (745, 224, 260, 0)  # int(round(260.5)) = 260: round half to even, like torchvision


@pytest.mark.skipif(not os.path.isdir(REF_IMAGES), reason="reference fixtures are only present in the build container")
def test_oracle_equals_pillow_on_reference_images():
    names = sorted(f for f in os.listdir(REF_IMAGES) if f.lower().endswith((".jpg", ".jpeg", ".png")))
    assert names
    for f in names:
        im = Image.open(os.path.join(REF_IMAGES, f))
        src = reader.decode_rgb_u8(im)
        assert np.array_equal(ro.clip_resize_crop_u8(src, 224), np.asarray(reader.clip_preprocess_u8(im, size=224))), f


def test_raw_batches_are_packed_with_offsets():
    """reader._collate with the decode-only preprocess: one packed byte buffer + offsets + (h, w) per image."""
    imgs = [_synthetic(h, w, i) for i, (h, w) in enumerate(((40, 60), (33, 21), (224, 224)))]
    samples = [{"image_raw": im, "image_filename": str(i)} for i, im in enumerate(imgs)]
    batch = reader._collate(samples, enable_image=True, enable_text=False, enable_metadata=False, pin=False)  # pylint: disable=protected-access
    raw = batch["image_raw"]
    assert raw["hw"].tolist() == [[40, 60], [33, 21], [224, 224]]
    assert raw["offsets"].tolist() == [0, 40 * 60 * 3, 40 * 60 * 3 + 33 * 21 * 3]
    flat = raw["pixels"].numpy()
    for im, o in zip(imgs, raw["offsets"]):
        assert np.array_equal(flat[o:o + im.size], im.reshape(-1))
    assert batch["image_filename"] == ["0", "1", "2"] and "image_tensor" not in batch


def test_decode_only_preprocess_keeps_the_reference_transform_for_modes_the_gpu_kernel_does_not_restate():
    """Pillow resamples palette / bilevel images with NEAREST and premultiplies alpha, and the reference resizes in the
    image's own mode before converting to RGB.  `DecodeRgbU8` therefore hands P / 1 / RGBA / LA / I;16 sources (and oversized
    ones) over as the HOST transform's ready crop -- which the GPU resample (here: its numpy restatement) passes through
    unchanged -- and only RGB / L sources as decoded pixels; either way the crop is Pillow's, byte for byte."""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (150, 201, 3), dtype=np.uint8)
    rgb = Image.fromarray(base)
    alpha = rng.integers(0, 256, (150, 201), dtype=np.uint8)
    cases = {
        "RGB": rgb, "L": rgb.convert("L"), "P": rgb.quantize(16), "1": rgb.convert("1"),
        "RGBA": Image.merge("RGBA", (*rgb.split(), Image.fromarray(alpha))), "LA": Image.merge("LA", (rgb.convert("L"), Image.fromarray(alpha))),
        "I;16": Image.fromarray((base[..., 0].astype(np.uint16) * 200)),
    }
    dec = reader.DecodeRgbU8(224)
    assert dec.raw_images and reader.decode_rgb_u8.raw_images
    for mode, im in cases.items():
        assert im.mode == mode
        want = np.asarray(reader.clip_preprocess_u8(im, size=224))
        src = dec(im)
        if mode in ("RGB", "L"):
            assert src.shape == (150, 201, 3)  # decoded pixels: the GPU does the geometry
        else:
            assert src.shape == (224, 224, 3) and np.array_equal(src, want), mode  # the host transform's crop
        assert np.array_equal(ro.clip_resize_crop_u8(src, 224), want), f"mode {mode}: the crop differs from the reference transform's"
    # converting first and resizing afterwards is NOT the reference's crop for these modes (which is why they take the host path)
    for mode in ("P", "RGBA"):
        first = ro.clip_resize_crop_u8(np.asarray(cases[mode].convert("RGB"), dtype=np.uint8), 224)
        assert not np.array_equal(first, np.asarray(reader.clip_preprocess_u8(cases[mode], size=224))), mode
    # oversized sources stay on the host as well (the kernel would refuse them and the partition would die)
    big = Image.fromarray(np.zeros((224 * 16 + 1, 224 * 16 + 8, 3), dtype=np.uint8))  # a 16 x down-scale of the shorter side
    assert reader.DecodeRgbU8(224)(big).shape == (224, 224, 3)
    assert reader.DecodeRgbU8(224, max_side=256)(Image.fromarray(base).resize((300, 100))).shape == (224, 224, 3)
