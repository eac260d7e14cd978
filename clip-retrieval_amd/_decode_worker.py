"""Decode worker process of the readers (`python -m clip_retrieval_amd._decode_worker`, started by reader._DecodePool).

The counterpart of the reference's DataLoader worker processes (clip_retrieval/clip_inference/reader.py:184-205): JPEG decode
+ the CLIP transform + tokenisation of chunks of raw samples, outside the parent's GIL.  A plain child process with a
length-prefixed pickle protocol on its stdin / stdout: nothing of the parent is forked (it may hold a HIP context and
threads) and the parent's main script is not re-imported (multiprocessing's spawn / forkserver children do that).
"""
import os
import pickle
import struct
import sys


def main():
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr  # the "Failed to load image ..." notes must not land in the reply stream

    def recv():
        hdr = inp.read(8)
        if len(hdr) < 8:
            return None
        return pickle.loads(inp.read(struct.unpack("<Q", hdr)[0]))

    def send(obj):
        blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
        out.write(struct.pack("<Q", len(blob)))
        out.write(blob)
        out.flush()

    try:
        from clip_retrieval_amd.reader import _decode_sample  # pylint: disable=import-outside-toplevel

        args = recv()  # (preprocess, tokenizer, enable_image, enable_text, enable_metadata)
    except Exception as e:  # pylint: disable=broad-except
        send(("error", repr(e)))
        return
    arena, arena_np = None, None
    spec = os.environ.get("CLIPX_DECODE_ARENA")
    if spec:
        try:
            import mmap  # pylint: disable=import-outside-toplevel

            import numpy as np  # pylint: disable=import-outside-toplevel

            fd, size = (int(v) for v in spec.split(":"))
            arena = mmap.mmap(fd, size)
            arena_np = np.frombuffer(arena, dtype=np.uint8)
        except (OSError, ValueError):
            arena = None
    send(("ok", None))
    while True:
        raws = recv()
        if raws is None:
            return
        try:
            outs = [_decode_sample(r, *args) for r in raws]
            if arena is not None:
                # pixels go back through the shared arena; what does not fit (a chunk of very large sources) is pickled as before
                off = 0
                for o in outs:
                    if o is None:
                        continue
                    for k in ("image_tensor", "image_raw"):
                        a = o.get(k)
                        if a is None or not hasattr(a, "nbytes") or off + a.nbytes > arena_np.size:
                            continue
                        flat = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
                        arena_np[off:off + flat.size] = flat
                        o[k] = ("@arena", off, a.shape, a.dtype.str)
                        off += (flat.size + 63) & ~63
            send(("ok", outs))
        except Exception as e:  # pylint: disable=broad-except
            send(("error", repr(e)))


if __name__ == "__main__":
    main()
