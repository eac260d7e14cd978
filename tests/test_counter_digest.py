"""The committed counter summaries (profiles/traffic.json, profiles/mfma_busy.json) are quoted by bench.py only when they were measured
on THIS tree's kernels: tools/source_digest.py stamps them with a sha256 over csrc/ + include/, bench.counters_match compares
(VERDICT r5 #9).  CPU-only: no GPU, no library call."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_digest_is_stable_and_sensitive(tmp_path, monkeypatch):
    import source_digest

    a = source_digest.source_digest()
    assert a == source_digest.source_digest() and len(a) == 16
    # a changed kernel source changes it: point the module at a copy of the tree with one byte appended
    import shutil

    for sub in ("clip-retrieval_amd/csrc", "include"):
        os.makedirs(tmp_path / sub, exist_ok=True)
        for f in os.listdir(os.path.join(ROOT, sub)):
            p = os.path.join(ROOT, sub, f)
            if os.path.isfile(p) and (f.endswith((".hip", ".h")) or f == "Makefile"):
                shutil.copy(p, tmp_path / sub / f)
    monkeypatch.setattr(source_digest, "ROOT", str(tmp_path))
    assert source_digest.source_digest() == a
    with open(tmp_path / "clip-retrieval_amd/csrc/knn_kernels.hip", "a") as fh:
        fh.write("\n")
    assert source_digest.source_digest() != a


def test_bench_ignores_counters_of_another_tree():
    import bench
    import source_digest

    assert bench.counters_match({"source_digest": source_digest.source_digest()})
    assert not bench.counters_match({"source_digest": "0" * 16})
    assert not bench.counters_match({})  # an unstamped file (rounds 1 - 5)


def test_committed_counter_files_are_stamped():
    for name in ("traffic.json", "mfma_busy.json"):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            j = json.load(f)
        assert isinstance(j.get("source_digest"), str) and len(j["source_digest"]) == 16, name
