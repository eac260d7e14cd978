"""GPU resize + centre crop (csrc/preprocess.hip, clipx_resize_crop_u8_device) against Pillow -- the code the
reference's image transform runs on the host (reader.py:83,87) -- and against the numpy restatement in oracle/resample_oracle.py.
Byte work: the bar is bit equality."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

SIZES = [(224, 224), (256, 256), (300, 451), (451, 300), (97, 131), (131, 97), (224, 500), (500, 224), (32, 32), (17, 400),
         (400, 17), (1024, 768), (225, 223), (223, 225), (1, 1), (2, 3000), (1500, 2100), (640, 480), (224, 225)]


def _synthetic(h, w, seed):
    from PIL import Image

    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (max(2, h // 7), max(2, w // 7), 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR), dtype=np.int16)
    return np.clip(img + rng.integers(-40, 41, img.shape), 0, 255).astype(np.uint8)


def _pack(imgs):
    hw = np.asarray([im.shape[:2] for im in imgs], dtype=np.int32)
    nb = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
    off = np.zeros(len(imgs), dtype=np.int64)
    np.cumsum(nb[:-1], out=off[1:])
    flat = np.concatenate([im.reshape(-1) for im in imgs])
    return flat, off, hw


def _resize_crop(lib, imgs, S):
    flat, off, hw = _pack(imgs)
    src = torch.from_numpy(flat).cuda()
    out = torch.empty((len(imgs), S, S, 3), dtype=torch.uint8, device="cuda")
    rc = lib.clipx_resize_crop_u8_device(0, C.c_void_p(src.data_ptr()), off.ctypes.data, hw.ctypes.data, len(imgs), S,
                                         C.c_void_p(out.data_ptr()), None)
    assert rc == 0, lib.clipx_last_error()
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("S", [224, 64, 336])
def test_kernel_equals_pillow_bit_for_bit(lib, S):
    from PIL import Image

    from clip_retrieval_amd.reader import clip_preprocess_u8

    imgs = [_synthetic(h, w, 31 * h + w) for h, w in SIZES]
    got = _resize_crop(lib, imgs, S)
    for im, g in zip(imgs, got):
        want = np.asarray(clip_preprocess_u8(Image.fromarray(im), size=S))
        bad = int((g != want).sum())
        assert bad == 0, f"source {im.shape[:2]} -> {S}: {bad} bytes differ (max |diff| {np.abs(g.astype(int) - want.astype(int)).max()})"


def test_kernel_equals_the_oracle_and_saturates_like_pillow(lib):
    """Black / white checkerboards and single-pixel lines: the bicubic lobes overshoot below 0 and above 255, so both clip8 ends
    are hit in both passes."""
    from oracle.resample_oracle import clip_resize_crop_u8

    imgs = []
    for h, w, p in ((300, 300, 1), (300, 451, 2), (100, 90, 3), (640, 480, 16), (200, 260, 50)):
        yy, xx = np.mgrid[:h, :w]
        imgs.append(np.repeat((((yy // p + xx // p) & 1) * 255).astype(np.uint8)[:, :, None], 3, axis=2))
    got = _resize_crop(lib, imgs, 224)
    for im, g in zip(imgs, got):
        assert np.array_equal(g, clip_resize_crop_u8(im, 224)), im.shape
    for g in got[3:]:  # squares wider than the filter: flat black and white areas survive, the edges between them overshoot
        assert g.min() == 0 and g.max() == 255


def test_batches_and_repeat_calls_are_independent(lib):
    """One image alone == the same image inside a ragged batch; many calls in a row (the coefficient slots are a ring of 4
    guarded by events) give the same bytes every time."""
    imgs = [_synthetic(h, w, 7 * h + w) for h, w in SIZES[:12]]
    whole = _resize_crop(lib, imgs, 224)
    for rep in range(10):
        i = rep % len(imgs)
        assert np.array_equal(_resize_crop(lib, [imgs[i]], 224)[0], whole[i])
    assert np.array_equal(_resize_crop(lib, imgs[::-1], 224)[::-1], whole)


def test_unaligned_source_and_repeated_sizes(lib):
    """A packed source that does not start on a 16-byte boundary takes the byte-wise staging path; images of one size share
    one copy of the weight tables (host cache + per-call placement).  Same bytes as Pillow either way."""
    from PIL import Image

    from clip_retrieval_amd.reader import clip_preprocess_u8

    imgs = [_synthetic(h, w, 5 * i + h) for i, (h, w) in enumerate(((256, 256), (256, 256), (300, 451), (256, 256), (300, 451), (97, 131)))]
    flat, off, hw = _pack(imgs)
    buf = torch.zeros(flat.size + 64, dtype=torch.uint8, device="cuda")
    for shift in (0, 1, 7):
        buf[shift:shift + flat.size] = torch.from_numpy(flat).cuda()
        out = torch.empty((len(imgs), 224, 224, 3), dtype=torch.uint8, device="cuda")
        rc = lib.clipx_resize_crop_u8_device(0, C.c_void_p(buf.data_ptr() + shift), off.ctypes.data, hw.ctypes.data, len(imgs), 224,
                                             C.c_void_p(out.data_ptr()), None)
        assert rc == 0, lib.clipx_last_error()
        torch.cuda.synchronize()
        for im, g in zip(imgs, out.cpu().numpy()):
            assert np.array_equal(g, np.asarray(clip_preprocess_u8(Image.fromarray(im), size=224))), (shift, im.shape)


def test_bad_arguments_are_refused(lib):
    im = _synthetic(40, 50, 1)
    flat, off, hw = _pack([im])
    src = torch.from_numpy(flat).cuda()
    out = torch.empty((1, 224, 224, 3), dtype=torch.uint8, device="cuda")
    f = lib.clipx_resize_crop_u8_device
    assert f(0, None, off.ctypes.data, hw.ctypes.data, 1, 224, C.c_void_p(out.data_ptr()), None) != 0
    assert f(0, C.c_void_p(src.data_ptr()), off.ctypes.data, hw.ctypes.data, 1, 0, C.c_void_p(out.data_ptr()), None) != 0
    bad_hw = np.asarray([[0, 50]], dtype=np.int32)
    assert f(0, C.c_void_p(src.data_ptr()), off.ctypes.data, bad_hw.ctypes.data, 1, 224, C.c_void_p(out.data_ptr()), None) != 0
    assert f(0, C.c_void_p(src.data_ptr()), off.ctypes.data, hw.ctypes.data, 0, 224, C.c_void_p(out.data_ptr()), None) == 0  # empty batch


def test_encode_image_raw_equals_encode_image_on_pillow_crops():
    """The whole seam: decoded sources -> GPU resize / crop / normalise -> image tower == Pillow crops -> GPU normalise -> image
    tower, the same bytes out (the crops are the same bytes in)."""
    from PIL import Image

    from clip_retrieval_amd.encoder import ClipArch, ClipEncoder
    from clip_retrieval_amd.reader import _collate, clip_preprocess_u8
    from oracle.clip_oracle import ARCHS, HFClipOracle

    arch = ARCHS["tiny-B/32"]
    enc = ClipEncoder(ClipArch(**{k: getattr(arch, k) for k in ClipArch.__dataclass_fields__}), HFClipOracle(arch, seed=0).export_blob(), 0)
    imgs = [_synthetic(h, w, h + 13 * w) for h, w in SIZES[:13]]
    crops = np.stack([np.asarray(clip_preprocess_u8(Image.fromarray(im), size=arch.image_size)) for im in imgs])
    want = enc.encode_image(torch.from_numpy(crops))
    batch = _collate([{"image_raw": im, "image_filename": str(i)} for i, im in enumerate(imgs)], True, False, False, True)
    got = enc.encode_image_raw(batch["image_raw"])
    assert got.dtype == want.dtype and np.array_equal(got, want)
    enc.close()
