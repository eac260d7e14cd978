"""a9: the BPE tokenizer restated in clip-retrieval_amd/tokenizer.py, pinned against an INDEPENDENT implementation of
the same published algorithm (transformers.CLIPTokenizer, pure Python) on a synthetic merges file -- the real
bpe_simple_vocab_16e6.txt.gz lives in the `clip` wheel, which is not available offline -- plus hand-worked vectors."""
import collections
import gzip

import numpy as np
import pytest

CORPUS = ("a photo of a cat . a photo of a dog ! the quick brown fox jumps over the lazy dog 's back ; photos photographer "
          "photography cats dogs catalog dogma the theory there these those 12 345 l'été naïve café don't i'll we've "
          "running runner runs ran lower lowest newer newest wider widest").split()


def _train_merges(n_merges):
    """Toy BPE training (Sennrich et al.) over CORPUS in CLIP's symbol alphabet: deterministic, ties by pair order."""
    from clip_retrieval_amd.tokenizer import bytes_to_unicode

    b2u = bytes_to_unicode()
    words = collections.Counter()
    for w in CORPUS:
        sym = [b2u[b] for b in w.lower().encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for p in zip(w[:-1], w[1:]):
                pairs[p] += c
        if not pairs:
            break
        best = min(pairs, key=lambda p: (-pairs[p], p))
        merges.append(best)
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1])
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            new[tuple(out)] += c
        words = new
    return merges


@pytest.fixture(scope="module")
def toks(tmp_path_factory):
    from clip_retrieval_amd.tokenizer import SimpleTokenizer

    d = tmp_path_factory.mktemp("bpe")
    merges = _train_merges(200)
    assert len(merges) > 100
    body = "#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n"
    gz = d / "bpe_simple_vocab_16e6.txt.gz"
    with gzip.open(gz, "wt", encoding="utf-8") as f:
        f.write(body)
    ours = SimpleTokenizer(bpe_path=str(gz))
    assert ours.vocab_size == 512 + len(merges) + 2
    from transformers import CLIPTokenizer

    # transformers >= 5: backed by the Rust `tokenizers` BPE model (NFC + whitespace + lower-case normaliser, the same split
    # pattern, byte-level alphabet, "</w>" suffix) -- an implementation that shares no code with ours
    hf = CLIPTokenizer(vocab=dict(ours.encoder), merges=[tuple(m) for m in merges])
    return ours, hf, merges


TEXTS = ["a photo of a cat", "A Photo of the DOG's back!", "photographer's catalog, 12 345 runners...", "there   are\ttabs\nand  newlines",
         "l'été naïve café", "don't i'll we've they'd", "xyzzy qwertyuiop (unseen words)", "emoji 🐈 and ¿symbols? #@!", "", "the " * 100]


def test_matches_an_independent_clip_bpe_implementation(toks):
    ours, hf, _ = toks
    for t in TEXTS:
        want = hf(t, add_special_tokens=False)["input_ids"]
        assert ours.encode(t) == want, t
    # [SOT] ... [EOT] + zero padding, and truncate=True semantics (last kept position becomes EOT)
    arr = ours.tokenize_numpy(TEXTS, context_length=77)
    assert arr.shape == (len(TEXTS), 77) and arr.dtype == np.int64
    for row, t in zip(arr, TEXTS):
        ids = ours.encode(t)
        n = min(len(ids) + 2, 77)
        assert row[0] == ours.sot_token and row[n - 1] == ours.eot_token and (row[n:] == 0).all()
        assert list(row[1:n - 1]) == ids[:n - 2]
        assert row.argmax() == n - 1  # EOT is the largest id: the pooled position of the text tower (argmax(ids))
    with pytest.raises(RuntimeError):
        ours.tokenize_numpy(["the " * 100], truncate=False)


def test_hand_worked_vectors_and_cleaning(toks):
    ours, _, merges = toks
    from clip_retrieval_amd.tokenizer import bytes_to_unicode

    b2u = bytes_to_unicode()
    # a single unseen character: byte symbol + </w>  -> id 256 + position of the byte in the alphabet
    order = list(b2u.values())
    assert ours.encode("~") == [256 + order.index("~")]
    # the first learnt merge is a vocabulary entry right after the 512 byte symbols
    assert ours.encoder["".join(merges[0])] == 512
    # html entities are unescaped twice, whitespace collapsed, text lower-cased
    assert ours.encode("A &amp;amp; B") == ours.encode("a & b")
    assert ours.encode("  A   PHOTO\n") == ours.encode("a photo")
    # round trip
    assert ours.decode(ours.encode("a photo of a cat")).strip() == "a photo of a cat"
    # callable form returns a torch tensor the readers index with [0] (reader.py:83)
    t = ours(["a cat"])
    assert tuple(t.shape) == (1, 77) and int(t[0, 0]) == ours.sot_token


def test_real_vocabulary_geometry_when_the_file_is_present():
    """With the real merges file (CLIP_BPE_PATH) the published constants must hold; skipped offline."""
    import os

    from clip_retrieval_amd.tokenizer import SimpleTokenizer

    if not os.environ.get("CLIP_BPE_PATH"):
        pytest.skip("CLIP_BPE_PATH not set (the merges file is not available offline)")
    tok = SimpleTokenizer()
    assert tok.vocab_size == 49408 and tok.sot_token == 49406 and tok.eot_token == 49407
    assert tok.encode("a photo of a cat") == [320, 1125, 539, 320, 2368]


def test_missing_file_fails_loudly(tmp_path, monkeypatch):
    from clip_retrieval_amd.tokenizer import MissingTokenizer, SimpleTokenizer

    monkeypatch.delenv("CLIP_BPE_PATH", raising=False)
    with pytest.raises(FileNotFoundError):
        SimpleTokenizer(clip_cache_path=str(tmp_path))
    with pytest.raises(FileNotFoundError):
        MissingTokenizer(FileNotFoundError("x"))(["a"])
