#!/usr/bin/env python3
"""GPU resize + centre crop (csrc/preprocess.hip) measured: images/s and source GB/s of the kernel alone (inputs resident in HBM),
the host-side cost of Pillow's resize for the same images (what the reference's DataLoader workers spend per image,
reader.py:83,87), and the encode seam both ways on the named model (decoded sources -> GPU resize -> tower, vs Pillow crops ->
tower; host buffers, so PCIe-inclusive).

    python tools/resize_bench.py [--model ViT-B/32 --batch 256 --reps 10]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-B/32")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()

    import numpy as np
    import torch
    from PIL import Image

    from clip_retrieval_amd import load_library
    from clip_retrieval_amd.encoder import load_clip
    from clip_retrieval_amd.reader import _collate, clip_preprocess_u8

    lib = load_library()
    rng = np.random.default_rng(0)
    out = {"batch": a.batch, "kernel": [], "encode": []}
    S = 224
    for h, w in ((256, 256), (480, 640), (768, 1024), (1536, 2048)):
        base = rng.integers(0, 256, (h // 8, w // 8, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR))
        imgs = [np.ascontiguousarray(np.roll(img, i, axis=1)) for i in range(a.batch)]
        batch = _collate([{"image_raw": im, "image_filename": str(i)} for i, im in enumerate(imgs)], True, False, False, True)["image_raw"]
        src = batch["pixels"].cuda()
        dst = torch.empty((a.batch, S, S, 3), dtype=torch.uint8, device="cuda")
        off, hw = batch["offsets"], batch["hw"]
        st = torch.cuda.current_stream().cuda_stream

        def run():
            rc = lib.clipx_resize_crop_u8_device(0, C.c_void_p(src.data_ptr()), off.ctypes.data, hw.ctypes.data, a.batch, S,
                                                 C.c_void_p(dst.data_ptr()), C.c_void_p(st))
            assert rc == 0, lib.clipx_last_error()

        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.reps
        ms = e0.elapsed_time(e1) / a.reps
        t0 = time.perf_counter()
        n_pil = min(a.batch, 64)
        for im in imgs[:n_pil]:
            clip_preprocess_u8(Image.fromarray(im), size=S)
        pil_ms = (time.perf_counter() - t0) / n_pil * 1e3
        row = {"source": f"{h}x{w}", "ms_per_batch_gpu": round(ms, 3), "ms_per_batch_wall": round(wall * 1e3, 3),
               "images_per_s": round(a.batch / (ms * 1e-3)), "source_GBps": round(src.numel() / (ms * 1e-3) / 1e9, 1),
               "pillow_ms_per_image_1core": round(pil_ms, 3)}
        out["kernel"].append(row)
        print(row, flush=True)
        del src, dst

    model, _, _ = load_clip("random:" + a.model, warmup_batch_size=a.batch)
    enc = model._enc  # pylint: disable=protected-access
    for h, w in ((256, 256), (480, 640)):
        base = rng.integers(0, 256, (h // 8, w // 8, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR))
        imgs = [np.ascontiguousarray(np.roll(img, i, axis=1)) for i in range(a.batch)]
        raw = _collate([{"image_raw": im, "image_filename": str(i)} for i, im in enumerate(imgs)], True, False, False, True)["image_raw"]
        crops = torch.from_numpy(np.stack([np.asarray(clip_preprocess_u8(Image.fromarray(im), size=enc.arch.image_size)) for im in imgs])).pin_memory()
        a_out = enc.encode_image_raw(raw)
        b_out = enc.encode_image(crops)
        same = bool(np.array_equal(a_out, b_out))
        res = {}
        for name, fn in (("gpu_resize", lambda: enc.encode_image_raw(raw)), ("pillow_crops", lambda: enc.encode_image(crops))):
            fn()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                fn()
            res[name] = round(a.batch * a.reps / (time.perf_counter() - t0))
        row = {"model": a.model, "source": f"{h}x{w}", "images_per_s": res, "embeddings_identical": same}
        out["encode"].append(row)
        print(row, flush=True)
    print("RESIZE " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
