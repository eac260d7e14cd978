"""Digest of the kernel sources a counter file was measured on: tools/traffic_summary.py and
tools/mfma_busy_summary.py stamp their JSON with it, bench.py recomputes it and refuses counters of another tree (the GPU box has
no .git, so the stamp is a sha256 over csrc/ and include/ instead of a commit hash)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_digest():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "clip-retrieval_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "clip-retrieval_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(ROOT, "clip-retrieval_amd", "csrc", "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_digest())
