"""Generate tests/golden/reference_reader_tensors.npz from the REFERENCE-HELD fixtures
/root/reference/tests/test_clip_inference/test_tensors/{0..3}.pkl (build container only).

Those pickles are the reference reader's own `image_tensor` batches (float32 [2 or 1, 3, 224, 224]) for
test_tars/image1.tar + image2.tar at batch size 2, written by the reference's author (tests/test_clip_inference/playground.ipynb)
and consumed by its test_mapper.py:31-36.  They need no model weights (the only reference-held golden VECTORS on
the whole path).  They are loaded with a restricted unpickler (three allowed globals; the tensor storage goes through
torch.load(weights_only=True)), NOT with pickle.load.

The tensors themselves are 4.2 MB, so what is committed is (a) the uint8 crops they were made from, RECOVERED FROM THE REFERENCE
TENSORS THEMSELVES (u8 = round((t * std + mean) * 255): no code of this repository is involved), (b) the proof that the recovery is
lossless: torchvision's arithmetic ((u8 / 255 - mean) / std in f32) applied to (a) reproduces every reference tensor BIT FOR BIT
(asserted here, and the sha256 of each reference tensor's bytes is stored), (c) the file names in batch order.
tests/test_reader_reference_tensors.py then holds reader.clip_preprocess / WebdatasetReader (CPU, against the pickles directly
when /root/reference is present, against this fixture otherwise) and the device-side normalisation (GPU) to those bytes."""
import collections
import hashlib
import io
import os
import pickle

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests/test_clip_inference/test_tensors"
MEAN = np.asarray((0.48145466, 0.4578275, 0.40821073), dtype=np.float32)
STD = np.asarray((0.26862954, 0.26130258, 0.27577711), dtype=np.float32)


class RestrictedUnpickler(pickle.Unpickler):
    """Only what a pickled {str: list | torch.Tensor} needs; the raw storage bytes are parsed by torch.load(weights_only=True)."""

    def find_class(self, module, name):
        if (module, name) == ("torch._utils", "_rebuild_tensor_v2"):
            return torch._utils._rebuild_tensor_v2  # pylint: disable=protected-access
        if (module, name) == ("collections", "OrderedDict"):
            return collections.OrderedDict
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return lambda b: torch.load(io.BytesIO(b), weights_only=True)
        raise pickle.UnpicklingError(f"global {module}.{name} is not allowed")


def load_reference_batches(folder=REF):
    out = []
    for i in range(4):
        with open(os.path.join(folder, f"{i}.pkl"), "rb") as f:
            d = RestrictedUnpickler(f).load()
        assert {"image_filename", "image_tensor"} <= set(d) <= {"image_filename", "image_tensor", "__key__"}, set(d)
        t = d["image_tensor"]
        assert isinstance(t, torch.Tensor) and t.dtype == torch.float32 and tuple(t.shape[1:]) == (3, 224, 224)
        out.append((list(d["image_filename"]), t.contiguous().numpy()))
    return out


def torchvision_normalise(u8_nhwc):
    x = u8_nhwc.astype(np.float32) / np.float32(255.0)      # ToTensor
    x = (x - MEAN) / STD                                     # Normalize
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def main():
    names, crops, shas, sizes = [], [], [], []
    for fn, t in load_reference_batches():
        u8 = np.rint((t.transpose(0, 2, 3, 1).astype(np.float64) * STD + MEAN) * 255.0)
        assert u8.min() >= 0 and u8.max() <= 255
        u8 = u8.astype(np.uint8)
        back = torchvision_normalise(u8)
        assert back.tobytes() == t.tobytes(), "torchvision's arithmetic must reproduce the reference tensor bit for bit"
        names += fn
        crops.append(u8)
        sizes.append(len(fn))
        shas.append(np.frombuffer(hashlib.sha256(t.tobytes()).digest(), dtype=np.uint8))
    path = os.path.join(HERE, "reference_reader_tensors.npz")
    np.savez_compressed(path, image_filename=np.asarray(names), crops_u8=np.concatenate(crops), batch_sizes=np.asarray(sizes),
                        tensor_sha256=np.stack(shas))
    print("wrote", path, names, sizes, os.path.getsize(path))


if __name__ == "__main__":
    main()
