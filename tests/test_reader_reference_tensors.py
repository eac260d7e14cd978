"""The reader pinned to the only reference-held golden VECTORS on the encode path:
/root/reference/tests/test_clip_inference/test_tensors/{0..3}.pkl are the reference reader's own `image_tensor` batches for
test_tars/image1.tar + image2.tar (batch size 2, two loader workers; written by the reference's author, consumed by its
test_mapper.py:31-36).  They need no weights.  CPU: `WebdatasetReader` must yield the same keys, the same batch order and
composition, and BIT-IDENTICAL tensors (torchvision's arithmetic: x / 255, then (x - mean) / std).  The reference folder does not
travel to the GPU box, so a fixture DERIVED FROM THE PICKLES ALONE (tests/golden/make_golden_reader_tensors.py: the uint8 crops
recovered from the tensors + the sha256 of every tensor) is committed; the GPU test holds the device-side normalisation to it."""
import hashlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
REF_T = "/root/reference/tests/test_clip_inference"
GOLD = os.path.join(HERE, "golden", "reference_reader_tensors.npz")


def _fixture():
    g = np.load(GOLD)
    return [str(x) for x in g["image_filename"]], g["crops_u8"], [int(x) for x in g["batch_sizes"]], g["tensor_sha256"]


def _normalise(u8):
    from make_golden_reader_tensors import torchvision_normalise

    return torchvision_normalise(u8)


def test_fixture_reproduces_the_reference_tensors_it_was_made_from():
    """torchvision's arithmetic on the committed uint8 crops gives tensors whose sha256 are the reference pickles' (so the 0.8 MB
    fixture stands for the 4.2 MB of reference tensors, bit for bit) -- and, when the reference is present, against the pickles."""
    names, crops, sizes, shas = _fixture()
    assert sizes == [2, 2, 2, 1] and len(names) == 7 and crops.shape == (7, 224, 224, 3)
    o = 0
    for n, sha in zip(sizes, shas):
        assert hashlib.sha256(_normalise(crops[o:o + n]).tobytes()).digest() == bytes(sha)
        o += n
    if os.path.isdir(REF_T):
        from make_golden_reader_tensors import load_reference_batches

        o = 0
        for (fn, t), n in zip(load_reference_batches(), sizes):
            assert fn == names[o:o + n] and t.tobytes() == _normalise(crops[o:o + n]).tobytes()
            o += n


def test_restricted_unpickler_refuses_other_globals():
    import io
    import pickle

    from make_golden_reader_tensors import RestrictedUnpickler

    with pytest.raises(pickle.UnpicklingError):
        RestrictedUnpickler(io.BytesIO(pickle.dumps(os.getcwd))).load()


@pytest.mark.skipif(not os.path.isdir(REF_T), reason="the reference's test_tars are only in the build container")
@pytest.mark.parametrize("use_processes", [True, False])
def test_webdataset_reader_yields_the_reference_batches_bit_for_bit(use_processes):
    """tests/test_clip_inference/test_reader.py:9-61's WebdatasetReader call on image1.tar + image2.tar: keys, batch order and
    composition (the reference's two loader workers each batch their own shard: [1a 1b] [2a 2b] [1c 1d] [2c]), shapes, dtype and
    every float of `image_tensor` equal to the reference-held pickles -- through `clip_preprocess` (f32 on the host) and through
    the uint8 reader that leaves the normalisation to the GPU (its crops must be the bytes the reference normalised)."""
    from clip_retrieval_amd.reader import ClipTransform, WebdatasetReader, clip_preprocess, clip_preprocess_u8
    from clip_retrieval_amd.runner import Sampler

    names, crops, sizes, shas = _fixture()
    tars = [REF_T + "/test_tars/image1.tar", REF_T + "/test_tars/image2.tar"]
    for pre, as_u8 in ((clip_preprocess, False), (ClipTransform(224), False), (clip_preprocess_u8, True)):
        r = WebdatasetReader(Sampler(0, 1), pre, None, tars, 2, 2, enable_text=False, enable_image=True, enable_metadata=False)
        r.use_processes = use_processes
        batches = list(r)
        assert [b["image_tensor"].shape[0] for b in batches] == sizes
        o = 0
        for b, n, sha in zip(batches, sizes, shas):
            assert set(b) == {"image_filename", "image_tensor"} and list(b["image_filename"]) == names[o:o + n]
            t = b["image_tensor"].numpy()
            if as_u8:
                assert t.dtype == np.uint8 and t.shape == (n, 224, 224, 3) and np.array_equal(t, crops[o:o + n])
            else:
                assert t.dtype == np.float32 and t.shape == (n, 3, 224, 224)
                assert hashlib.sha256(np.ascontiguousarray(t).tobytes()).digest() == bytes(sha), f"batch at {o}: image_tensor differs from the reference's"
            o += n
    # one stream (num_prepro_workers <= 1, or reference_batch_order = False): the shards in order, batches across shard borders
    r = WebdatasetReader(Sampler(0, 1), clip_preprocess, None, tars, 2, 1, enable_text=False, enable_image=True, enable_metadata=False)
    got = [list(b["image_filename"]) for b in r]
    assert got == [["123_456", "208_495"], ["321_421", "389_535"], ["416_264", "456_123"], ["524_316"]]


@pytest.mark.skipif(not os.path.isdir(REF_T), reason="the reference's test_tars are only in the build container")
def test_reference_reader_test_batch_shapes_over_four_tars():
    """The reference's own assertion (test_reader.py:58-61): 4 tars, 2 output partitions, 2 loader workers, batch 2 ->
    [[2, 2, 2], [2, 2, 1]]."""
    from clip_retrieval_amd.reader import WebdatasetReader, clip_preprocess
    from clip_retrieval_amd.runner import Sampler

    tars = [REF_T + f"/test_tars/image{i}.tar" for i in (1, 2, 3, 4)]
    got = []
    for pid in range(2):
        r = WebdatasetReader(Sampler(pid, 2), clip_preprocess, None, tars, 2, 2, enable_text=False, enable_image=True, enable_metadata=False)
        got.append([b["image_tensor"].shape[0] for b in r])
    assert got == [[2, 2, 2], [2, 2, 1]]


@pytest.mark.gpu
def test_device_normalisation_reproduces_the_reference_tensors():
    """CLIPX_PIX_U8_NHWC (the /255, mean / std arithmetic inside the patch-gather kernel) against the reference reader's tensors:
    the encoder fed the raw uint8 crops must return the SAME BYTES as the encoder fed the reference's f32 `image_tensor` --
    the device arithmetic is torchvision's, operation for operation, so both inputs round to the same bf16 patches."""
    from clip_retrieval_amd.encoder import ARCHS, ClipEncoder, random_blob

    names, crops, sizes, shas = _fixture()
    ref_f32 = _normalise(crops)
    o = 0
    for n, sha in zip(sizes, shas):
        assert hashlib.sha256(ref_f32[o:o + n].tobytes()).digest() == bytes(sha)
        o += n
    arch = ARCHS["ViT-B/32"]
    enc = ClipEncoder(arch, random_blob(arch, seed=0), 0)
    a, b = enc.encode_image(crops), enc.encode_image(ref_f32)
    enc.close()
    assert a.dtype == np.float16 and a.shape == (7, arch.embed_dim)
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), "device-side normalisation differs from the reference reader's tensor"
    c = a.astype(np.float64) @ a.astype(np.float64).T
    assert c[~np.eye(7, dtype=bool)].max() < 0.9999  # seven different photographs
