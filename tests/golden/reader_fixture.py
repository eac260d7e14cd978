"""Deterministic image/caption/metadata folder shared by tests/golden/make_golden_reader.py (which feeds it to the
REFERENCE's FilesReader / Runner / NumpyWriter) and tests/test_host_logic.py (which feeds it to ours)."""
import io
import json
import os

import numpy as np

SIZES = [(123, 456), (208, 495), (321, 421), (389, 535), (416, 264), (456, 123), (524, 316), (64, 64), (300, 300)]


def jpeg_bytes(w, h, seed):
    from PIL import Image

    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), rng.integers(0, 255, (h, w))], -1).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="PNG")  # lossless: decoded pixels do not depend on the libjpeg build
    return buf.getvalue()


def make_folder(root):
    """9 PNG images + captions + json metadata, plus one corrupt image WITH caption (must be skipped: reader.py:100-104)."""
    os.makedirs(root, exist_ok=True)
    for i, (w, h) in enumerate(SIZES):
        stem = f"{i:03d}_{w}x{h}"
        with open(os.path.join(root, stem + ".png"), "wb") as f:
            f.write(jpeg_bytes(w, h, i))
        with open(os.path.join(root, stem + ".txt"), "w", encoding="utf-8") as f:
            f.write(f"a synthetic gradient number {i} of size {w} by {h}")
        with open(os.path.join(root, stem + ".json"), "w", encoding="utf-8") as f:
            json.dump({"url": f"http://example.org/{i}", "width": w, "height": h}, f)
    with open(os.path.join(root, "004_broken.png"), "wb") as f:
        f.write(b"this is not an image")
    with open(os.path.join(root, "004_broken.txt"), "w", encoding="utf-8") as f:
        f.write("caption of a broken image")
    with open(os.path.join(root, "004_broken.json"), "w", encoding="utf-8") as f:
        json.dump({"url": "broken"}, f)
    return root


class StubMapper:
    """Embeddings that are a pure function of the batch, so files written downstream are comparable bit for bit."""

    def __call__(self, item):
        import torch

        x = item["image_tensor"]
        img = torch.stack([x.mean(dim=(1, 2, 3)), x.amax(dim=(1, 2, 3)), x[:, 0].mean(dim=(1, 2)), x[:, 2].std(dim=(1, 2))], 1)
        return {"image_embs": img.numpy().astype(np.float16), "text_embs": None, "image_filename": item["image_filename"],
                "text": None, "metadata": None}


class ListLogger:
    def __init__(self, i):
        self.i, self.records = i, []

    def start(self):
        pass

    def end(self):
        pass

    def __call__(self, stats):
        self.records.append(stats)
