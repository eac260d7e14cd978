"""NumpyWriter with the reference's on-disk format (adjacent to the hot path: our index loader reads it).

Format kept byte-compatible with clip_retrieval/clip_inference/writer.py:58-112:
  <out>/img_emb/img_emb_<N>.npy, <out>/text_emb/text_emb_<N>.npy   np.save of the concatenated fp16 batches
  <out>/metadata/metadata_<N>.parquet                               image_path / caption (+ json columns)
  N = partition id zero-filled to int(log10(output_partition_count)) + 1 digits (writer.py:22,67)
Nothing is written before flush(); an empty partition writes nothing (writer.py:108-112).
"""

import io
import json
import math

import fsspec
import numpy as np


class NumpyWriter:
    """Buffers mapper outputs and writes one .npy (+ parquet) per partition at flush()."""

    def __init__(self, partition_id, output_folder, enable_text, enable_image, enable_metadata, output_partition_count):
        self.enable_text, self.enable_image, self.enable_metadata = enable_text, enable_image, enable_metadata
        self.fs, self.root = fsspec.core.url_to_fs(output_folder)
        self.tag = str(partition_id).zfill(int(math.log10(output_partition_count)) + 1)
        if enable_image:
            self.fs.makedirs(self.root + "/img_emb", exist_ok=True)
        if enable_text:
            self.fs.makedirs(self.root + "/text_emb", exist_ok=True)
        self.fs.makedirs(self.root + "/metadata", exist_ok=True)
        self._reset()

    def _reset(self):
        self.img, self.txt, self.names, self.captions, self.meta, self.rows = [], [], [], [], [], 0

    def __call__(self, sample):
        self.rows += (sample["image_embs"] if self.enable_image else sample["text_embs"]).shape[0]
        if self.enable_image:
            self.img.append(sample["image_embs"])
            self.names.extend(sample["image_filename"])
        if self.enable_text:
            self.txt.append(sample["text_embs"])
            self.captions.extend(sample["text"])
        if self.enable_metadata:
            self.meta.extend(sample["metadata"])

    def _save_npy(self, path, parts):
        buf = io.BytesIO()
        np.save(buf, np.concatenate(parts))
        with self.fs.open(path, "wb") as f:
            f.write(buf.getbuffer())

    def flush(self):
        if self.rows == 0:
            return
        import pandas as pd  # pylint: disable=import-outside-toplevel

        columns = {}
        if self.enable_image:
            self._save_npy(f"{self.root}/img_emb/img_emb_{self.tag}.npy", self.img)
            columns["image_path"] = self.names
        if self.enable_text:
            self._save_npy(f"{self.root}/text_emb/text_emb_{self.tag}.npy", self.txt)
            columns["caption"] = self.captions
        if self.enable_metadata:
            columns["metadata"] = self.meta
        df = pd.DataFrame(columns)
        if self.enable_metadata:
            parsed = pd.json_normalize(df["metadata"].apply(json.loads))
            parsed = parsed.drop(columns=list({"caption", "metadata", "image_path"} & set(parsed.keys())))
            df = df.join(parsed).drop(columns=["metadata"])
        with self.fs.open(f"{self.root}/metadata/metadata_{self.tag}.parquet", "wb") as f:
            df.to_parquet(f)
        self._reset()
