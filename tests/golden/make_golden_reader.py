"""Generate tests/golden/reference_reader_runner.json by running the REFERENCE's own FilesReader, Runner and NumpyWriter
(clip_retrieval/clip_inference/{reader,runner,writer}.py, loaded by file path, executed unmodified; build container only)
over the deterministic folder of tests/golden/reader_fixture.py, with this repository's `clip_preprocess` /
`HashTokenizer` plugged in as the `preprocess` / `tokenizer` the reference takes from the model package, and a stub mapper.
What is recorded: per output partition the sequence of batches (file names, captions, metadata, token rows, SHA-256 of
the image tensor bytes) and the bytes / rows of every file the reference writer produced.
Usage: python tests/golden/make_golden_reader.py"""
import hashlib
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/clip_retrieval/clip_inference"

from reader_fixture import ListLogger, StubMapper, make_folder  # noqa: E402


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    import torch

    from clip_retrieval_amd.reader import HashTokenizer, clip_preprocess

    reader, runner, writer = load("reader"), load("runner"), load("writer")
    preprocess = lambda im: torch.from_numpy(clip_preprocess(im))  # noqa: E731
    golden = {"partitions": [], "files": {}}
    with tempfile.TemporaryDirectory() as tmp:
        folder = make_folder(os.path.join(tmp, "in"))
        out = os.path.join(tmp, "out")
        count, batch_size = 2, 3
        seen = {}

        class Recording:
            """wraps the reference reader to record what it yields, then hands the batch on unchanged"""

            def __init__(self, sampler):
                # image-only, the configuration the reference's own tests run (tests/test_clip_inference/test_reader.py);
                # with enable_text=True this snapshot of folder_to_keys keys the folder by "<name>.txt" and its
                # ImageDataset then fails on image_files[key] (reader.py:36-49,99)
                self.inner = reader.FilesReader(sampler, preprocess, HashTokenizer(), folder, batch_size, 0,
                                                enable_text=False, enable_image=True, enable_metadata=False)
                self.pid = sampler.output_partition_id

            def __iter__(self):
                for b in self.inner:
                    seen.setdefault(self.pid, []).append({
                        "image_filename": [os.path.basename(p) for p in b["image_filename"]],
                        "keys": sorted(b.keys()),
                        "image_shape": list(b["image_tensor"].shape), "image_dtype": str(b["image_tensor"].dtype),
                        "image_sha256": hashlib.sha256(b["image_tensor"].numpy().tobytes()).hexdigest()})
                    yield b

        r = runner.Runner(reader_builder=Recording, mapper_builder=StubMapper,
                          writer_builder=lambda i: writer.NumpyWriter(i, out, False, True, False, count),
                          logger_builder=ListLogger, output_partition_count=count)
        for i in range(count):
            r(i)
        golden["partitions"] = [seen.get(i, []) for i in range(count)]
        for root, _, names in os.walk(out):
            for name in sorted(names):
                p = os.path.join(root, name)
                rel = os.path.relpath(p, out)
                if name.endswith(".npy"):
                    golden["files"][rel] = hashlib.sha256(open(p, "rb").read()).hexdigest()
                else:
                    import pandas as pd

                    df = pd.read_parquet(p)
                    golden["files"][rel] = {"columns": list(df.columns), "rows": json.loads(df.to_json(orient="records"))}
    with open(os.path.join(HERE, "reference_reader_runner.json"), "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print("wrote reference_reader_runner.json:", [len(p) for p in golden["partitions"]], "batches;", sorted(golden["files"]))


if __name__ == "__main__":
    main()
