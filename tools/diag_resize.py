#!/usr/bin/env python3
"""Where the GPU resize differs from Pillow (debugging aid): per image the mismatch count and the rows / columns / channels hit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from PIL import Image

import test_preprocess_gpu as T
from clip_retrieval_amd import load_library
from clip_retrieval_amd.reader import clip_preprocess_u8

lib = load_library()
S = 224
for sizes in (T.SIZES[:1], T.SIZES[:4], T.SIZES):
    imgs = [T._synthetic(h, w, 31 * h + w) for h, w in sizes]
    got = T._resize_crop(lib, imgs, S)
    print("batch of", len(sizes))
    for im, g in zip(imgs, got):
        want = np.asarray(clip_preprocess_u8(Image.fromarray(im), size=S))
        bad = g != want
        if bad.any():
            rows = np.nonzero(bad.any(axis=(1, 2)))[0]
            cols = np.nonzero(bad.any(axis=(0, 2)))[0]
            print(" ", im.shape[:2], int(bad.sum()), "rows", rows[:20], "n", len(rows), "cols", cols[:20], "n", len(cols), "ch", bad.sum(axis=(0, 1)))
            y, x, c = np.argwhere(bad)[0]
            print("    first", (y, x, c), "got", g[y, x], "want", want[y, x], "row got", g[y, x:x + 6, 0], "want", want[y, x:x + 6, 0])
        else:
            print(" ", im.shape[:2], "ok")
