"""CLIP byte-pair tokenizer for the text-encoder boundary (host side).

Stands behind ``utils/clip_tokenizer.py::SimpleTokenizer.encode`` (:274-280), which the reference calls from
``OpenCLIP._encode_text`` (models/objects.py:136-138).  Independent implementation of the published
OpenAI/open_clip algorithm: clean (html-unescape twice, collapse whitespace, lower) -> split into
contractions / letter runs / single digits / punctuation runs -> bytes-to-unicode -> greedy lowest-rank
BPE merges -> ids.  The merge table is the public ``bpe_simple_vocab_16e6.txt.gz`` that clearcam ships in
``utils/``; its path comes from ``bpe_path``, ``$CLEARCAM_BPE_VOCAB`` or the usual locations.
"""
from __future__ import annotations

import gzip
import html
import os
import unicodedata
from typing import Dict, List, Optional, Tuple

SOT_ID, EOT_ID, CONTEXT = 49406, 49407, 77
_CONTRACTIONS = ("'s", "'t", "'re", "'ve", "'m", "'ll", "'d")
_N_MERGES = 49152 - 256 - 2


def find_vocab(bpe_path: Optional[str] = None) -> str:
    here = os.path.dirname(os.path.abspath(__file__))
    cands = [bpe_path, os.environ.get("CLEARCAM_BPE_VOCAB"),
             os.path.join(here, "assets", "bpe_simple_vocab_16e6.txt.gz"),
             os.path.join(os.getcwd(), "utils", "bpe_simple_vocab_16e6.txt.gz")]      # inside a clearcam checkout
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise FileNotFoundError("bpe_simple_vocab_16e6.txt.gz not found: pass bpe_path= or set CLEARCAM_BPE_VOCAB "
                            "(clearcam ships it as utils/bpe_simple_vocab_16e6.txt.gz)")


def _byte_alphabet() -> Dict[int, str]:
    """GPT-2 style reversible byte -> printable-unicode map."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _kind(ch: str) -> str:
    cat = unicodedata.category(ch)
    if cat[0] == "L":
        return "L"
    if cat[0] == "N":
        return "N"
    return "S" if ch.isspace() else "P"


def split_words(text: str, specials=("<start_of_text>", "<end_of_text>")) -> List[str]:
    """The CLIP pre-tokenisation pattern without the `regex` package:
    special | 's|'t|'re|'ve|'m|'ll|'d | letters+ | one digit | other non-space run."""
    out, i, n = [], 0, len(text)
    while i < n:
        hit = next((s for s in specials if text.startswith(s, i)), None) or \
            next((c for c in _CONTRACTIONS if text.startswith(c, i)), None)
        if hit:
            out.append(hit)
            i += len(hit)
            continue
        k = _kind(text[i])
        if k == "S":
            i += 1
        elif k == "N":
            out.append(text[i])
            i += 1
        else:
            j = i + 1
            while j < n and _kind(text[j]) == k:
                j += 1
            out.append(text[i:j])
            i = j
    return out


class SimpleTokenizer:
    def __init__(self, bpe_path: Optional[str] = None, context_length: int = CONTEXT, sparse_merges: Optional[Dict[Tuple[str, str], int]] = None):
        """bpe_path / $CLEARCAM_BPE_VOCAB / the usual locations name clearcam's ``bpe_simple_vocab_16e6.txt.gz``.
        sparse_merges (tests): {(left, right): rank} for a SUBSET of the published merge table, ranks as published — enough to
        tokenise the strings it was cut for exactly (every merge those strings can ever look up must be present)."""
        if sparse_merges is not None:
            merges: List[Tuple[str, str]] = [("\x00unused", str(i)) for i in range(_N_MERGES)]      # placeholders keep the id layout
            for pair, rank in sparse_merges.items():
                merges[rank] = tuple(pair)
        else:
            lines = gzip.open(find_vocab(bpe_path)).read().decode("utf-8").split("\n")
            merges = [tuple(l.split()) for l in lines[1:_N_MERGES + 1]]
        self.byte_map = _byte_alphabet()
        alphabet = list(self.byte_map.values())
        # id order of the published vocab: bytes in keep-then-extra order, the same with </w>, merges, specials
        order = sorted(self.byte_map, key=lambda b: (ord(self.byte_map[b]) >= 256, ord(self.byte_map[b]) if ord(self.byte_map[b]) >= 256 else b))
        alphabet = [self.byte_map[b] for b in order]
        vocab = alphabet + [c + "</w>" for c in alphabet] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot_token_id, self.eot_token_id = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]
        self.vocab_size = len(vocab)
        self.context_length = context_length
        self._cache: Dict[str, List[int]] = {}

    @staticmethod
    def clean(text: str) -> str:
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split()).strip().lower()

    def _merge(self, word: str) -> List[int]:
        if word in self._cache:
            return self._cache[word]
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        while len(parts) > 1:
            best, at = None, -1
            for i in range(len(parts) - 1):
                r = self.rank.get((parts[i], parts[i + 1]))
                if r is not None and (best is None or r < best):
                    best, at = r, i
            if best is None:
                break
            a, b = parts[at], parts[at + 1]
            merged, i = [], 0
            while i < len(parts):                       # merge every occurrence of the winning pair, left to right
                if i < len(parts) - 1 and parts[i] == a and parts[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        ids = [self.encoder[p] for p in parts]
        self._cache[word] = ids
        return ids

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for w in split_words(self.clean(text)):
            if w in ("<start_of_text>", "<end_of_text>"):
                ids.append(self.encoder[w])
                continue
            ids.extend(self._merge("".join(self.byte_map[b] for b in w.encode("utf-8"))))
        return ids

    def tokens_for_model(self, text: str):
        """``OpenCLIP._encode_text`` :136-140: [SOT] + encode(text) + [EOT], zero padded to 77 (never truncated)."""
        import numpy as np
        t = [self.sot_token_id] + self.encode(text) + [self.eot_token_id]
        if len(t) < self.context_length:
            t += [0] * (self.context_length - len(t))
        return np.asarray([t], dtype=np.int32)
