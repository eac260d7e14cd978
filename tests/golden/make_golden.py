"""Generate tests/golden/*.json|*.npy from the REFERENCE's own code, run in the build container only.

The reference package cannot be imported as a whole here (clip_retrieval/__init__.py pulls faiss and
flask_restful), but two of its modules on the hot path's edges depend only on stdlib / numpy /
pandas / fsspec and are loaded by file path:
    clip_retrieval/clip_inference/runner.py  -> Sampler
    clip_retrieval/clip_inference/writer.py  -> NumpyWriter (npy bytes, file names)
`get_task_list` lives in slurm_worker.py, whose module imports `fire` (absent); its expected outputs are
taken verbatim from the reference's own test (tests/test_clip_inference/test_get_tasks.py:12-16,31-35).
Usage:  python tests/golden/make_golden.py   (needs /root/reference; the outputs are committed)
"""
import hashlib
import importlib.util
import json
import os
import tempfile

import numpy as np

REF = "/root/reference/clip_retrieval/clip_inference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    runner, writer = load("runner"), load("writer")
    golden = {"sampler": [], "get_task_list": [], "writer": []}
    for n, count in [(7, 2), (11, 3), (4, 4), (0, 2), (5, 1), (3, 5)]:
        items = [f"k{i:03d}" for i in range(n)]
        for pid in range(count):
            golden["sampler"].append({"n": n, "count": count, "id": pid, "out": runner.Sampler(pid, count)(items)})
    # reference test expectations (tests/test_clip_inference/test_get_tasks.py)
    golden["get_task_list"] = [
        {"num_tasks": 11, "world_size": 3, "out": [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10]]},
        {"num_tasks": 9, "world_size": 3, "out": [[0, 1, 2], [3, 4, 5], [6, 7, 8]]},
    ]
    rng = np.random.default_rng(0)
    for pid, pcount, rows in [(0, 2, [2, 2]), (3, 12, [3]), (7, 1000, [1, 2, 1])]:
        with tempfile.TemporaryDirectory() as tmp:
            w = writer.NumpyWriter(pid, tmp, True, True, True, pcount)
            batches = []
            for i, r in enumerate(rows):
                img = rng.standard_normal((r, 8)).astype(np.float16)
                txt = rng.standard_normal((r, 8)).astype(np.float16)
                b = {"image_embs": img, "text_embs": txt, "image_filename": [f"f{i}_{j}.jpg" for j in range(r)],
                     "text": [f"caption {i} {j}" for j in range(r)],
                     "metadata": [json.dumps({"url": f"http://x/{i}/{j}", "caption": "dup", "n": j}) for j in range(r)]}
                batches.append({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in b.items()})
                w(b)
            w.flush()
            files = {}
            for root, _, names in os.walk(tmp):
                for name in names:
                    p = os.path.join(root, name)
                    rel = os.path.relpath(p, tmp)
                    if name.endswith(".npy"):
                        files[rel] = hashlib.sha256(open(p, "rb").read()).hexdigest()
                    else:
                        import pandas as pd

                        df = pd.read_parquet(p)
                        files[rel] = {"columns": list(df.columns), "rows": json.loads(df.to_json(orient="records"))}
            golden["writer"].append({"partition_id": pid, "partition_count": pcount, "batches": batches, "files": files})
    with open(os.path.join(HERE, "reference_host_logic.json"), "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "reference_host_logic.json"))


if __name__ == "__main__":
    main()
