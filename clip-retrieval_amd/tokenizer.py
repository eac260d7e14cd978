"""CLIP's byte-pair-encoding tokenizer: str -> int [77].

The reference obtains it as the third value of `load_clip` (clip_retrieval/clip_inference/worker.py:52-57) and calls it as
`tokenizer([caption])[0]` in the readers (reader.py:83,87,144-145) and `tokenizer([text])` in the service (clip_back.py:227).
The implementation lives in third-party wheels that are not installed here (`clip.simple_tokenizer.SimpleTokenizer`,
re-exported by open_clip; all_clip calls it with truncate=True), so this is restated from OpenAI CLIP's `simple_tokenizer.py`
(MIT licence; bit-compatibility with its ids forces the same steps):

  text -> html.unescape twice, strip (ftfy.fix_text first when the `ftfy` package is importable; it is optional in CLIP
          too) -> collapse whitespace -> lower-case
       -> split with the pattern  <|startoftext|> | <|endoftext|> | 's | 't | 're | 've | 'm | 'll | 'd | letters+ | one digit |
          other-non-space+
       -> every piece: UTF-8 bytes mapped to printable code points (GPT-2 bytes_to_unicode), last symbol + "</w>", then the
          lowest-ranked adjacent pair is merged repeatedly (ranks = line order of the merges file)
       -> ids: vocabulary = 256 byte symbols, the same 256 with "</w>", one entry per merge, <|startoftext|>, <|endoftext|>
       -> [SOT] ids [EOT], zero padded to context_length; longer inputs are truncated and end with EOT (truncate=True).

The merges file is `bpe_simple_vocab_16e6.txt.gz` of the CLIP repository (first line is a header; CLIP uses lines
1 .. 49152-256-2 = 48 894 merges -> vocabulary 49 408, SOT = 49406, EOT = 49407).  It is NOT bundled (no network here):
pass `bpe_path`, set CLIP_BPE_PATH, or put the file next to the checkpoint in `clip_cache_path`.  Without it construction
fails loudly -- there is no stand-in on the product path (`reader.HashTokenizer` exists for synthetic benchmarks only).
Parity pin: tests/test_tokenizer.py checks this class id-for-id against `transformers.CLIPTokenizer` (an independent
implementation of the same algorithm) on a synthetic merges file, plus hand-worked vectors.
"""

import gzip
import html
import os
from functools import lru_cache

import numpy as np

BPE_FILE_NAME = "bpe_simple_vocab_16e6.txt.gz"
CLIP_N_MERGES = 49152 - 256 - 2


@lru_cache()
def bytes_to_unicode():
    """byte value -> printable unicode character (GPT-2's reversible byte alphabet)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def _pairs(word):
    return set(zip(word[:-1], word[1:]))


def basic_clean(text):
    try:
        import ftfy  # pylint: disable=import-outside-toplevel

        text = ftfy.fix_text(text)
    except ImportError:
        pass
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text):
    return " ".join(text.split()).strip()


def find_bpe_file(bpe_path=None, clip_cache_path=None):
    cands = []
    if bpe_path:
        cands.append(bpe_path)
    if os.environ.get("CLIP_BPE_PATH"):
        cands.append(os.environ["CLIP_BPE_PATH"])
    if clip_cache_path:
        cands.append(os.path.join(clip_cache_path if os.path.isdir(clip_cache_path) else os.path.dirname(clip_cache_path), BPE_FILE_NAME))
    for c in cands:
        if os.path.isfile(c):
            return c
    raise FileNotFoundError(
        f"CLIP BPE merges file not found (looked at {cands or 'nothing'}): pass bpe_path=, set CLIP_BPE_PATH or put "
        f"{BPE_FILE_NAME} (from the CLIP repository) into clip_cache_path.  It is not bundled and there is no network here.")


class SimpleTokenizer:
    """CLIP BPE tokenizer; `tok(texts, context_length=77)` -> torch int tensor [n, context_length] (numpy with as_numpy)."""

    def __init__(self, bpe_path=None, clip_cache_path=None, n_merges=CLIP_N_MERGES, context_length=77):
        import regex  # pylint: disable=import-outside-toplevel

        path = find_bpe_file(bpe_path, clip_cache_path)
        opener = gzip.open if path.endswith(".gz") else open
        with opener(path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(m.split()) for m in lines[1:n_merges + 1] if len(m.split()) == 2]
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab.extend("".join(m) for m in merges)
        vocab.extend(["<|startoftext|>", "<|endoftext|>"])
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = regex.compile(
            r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)
        self.vocab_size = len(self.encoder)
        self.sot_token = self.encoder["<|startoftext|>"]
        self.eot_token = self.encoder["<|endoftext|>"]
        self.context_length = context_length

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new_word, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new_word.extend(word[i:])
                    break
                new_word.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new_word.append(first + second)
                    i += 2
                else:
                    new_word.append(word[i])
                    i += 1
            word = tuple(new_word)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text):
        ids = []
        text = whitespace_clean(basic_clean(text)).lower()
        for token in self.pat.findall(text):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        return ids

    def decode(self, tokens):
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def tokenize_numpy(self, texts, context_length=None, truncate=True):
        if isinstance(texts, str):
            texts = [texts]
        L = context_length or self.context_length
        out = np.zeros((len(texts), L), dtype=np.int64)
        for i, t in enumerate(texts):
            ids = [self.sot_token] + self.encode(t) + [self.eot_token]
            if len(ids) > L:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {L}")
                ids = ids[:L]
                ids[-1] = self.eot_token
            out[i, : len(ids)] = ids
        return out

    def __call__(self, texts, context_length=None, truncate=True):
        import torch  # pylint: disable=import-outside-toplevel

        return torch.from_numpy(self.tokenize_numpy(texts, context_length, truncate))


class MissingTokenizer:
    """What `load_clip` returns as the tokenizer when no merges file can be found: image-only pipelines never call it;
    any call raises the FileNotFoundError that names the places that were searched (fails loudly at the point of use)."""

    def __init__(self, error):
        self._error = error

    def __call__(self, *args, **kwargs):
        raise self._error
