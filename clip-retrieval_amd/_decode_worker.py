"""Decode worker process of the readers (`python -m clip_retrieval_amd._decode_worker`, started by reader._DecodePool).

The counterpart of the reference's DataLoader worker processes (clip_retrieval/clip_inference/reader.py:184-205): JPEG decode
+ the CLIP transform + tokenisation of chunks of raw samples, outside the parent's GIL.  A plain child process with a
length-prefixed pickle protocol on its stdin / stdout: nothing of the parent is forked (it may hold a HIP context and
threads) and the parent's main script is not re-imported (multiprocessing's spawn / forkserver children do that).
"""
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
    send(("ok", None))
    while True:
        raws = recv()
        if raws is None:
            return
        try:
            send(("ok", [_decode_sample(r, *args) for r in raws]))
        except Exception as e:  # pylint: disable=broad-except
            send(("error", repr(e)))


if __name__ == "__main__":
    main()
