"""Byte-level BPE tokenizer for the reference's text tower (row T1's input side).

The reference tokenises gene sentences ("top-N gene symbols joined by spaces",
src/spaglam_preproc/core/gene_encoder.py:5-29) with open_clip's ``SimpleTokenizer``
(src/open_clip/tokenizer.py:127-265): lower-cased text, a regex pre-split, GPT-2 style reversible byte -> unicode
mapping, greedy lowest-rank pair merging, ``<start_of_text>`` / ``<end_of_text>`` framing, zero padding to the context
length and truncation that keeps the end token.  This module is an independent implementation of that published
algorithm.  The merge table itself is data, not code: OpenAI CLIP's ``bpe_simple_vocab_16e6.txt.gz`` (shipped with every
open_clip install) is read from ``vocab_path`` / ``$SC_BPE_VOCAB``; it is not redistributed here.

Cleaning: ``html.unescape`` twice, whitespace collapse, lower-case -- the reference's ``clean='lower'``.  Its
``ftfy.fix_text`` step (mojibake repair) is applied only if ftfy is importable; gene sentences are ASCII."""
from __future__ import annotations

import gzip
import html
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import regex
import torch

SOT, EOT = "<start_of_text>", "<end_of_text>"
N_MERGES = 49152 - 256 - 2            # merges kept by CLIP: vocabulary 49408 = 2 * 256 + 48894 + 2 specials


def _byte_symbols() -> Dict[int, str]:
    """Reversible byte -> printable unicode symbol table (GPT-2 / CLIP): printable latin-1 bytes map to themselves,
    the remaining 68 bytes to code points 256, 257, ... in byte order."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _vocab_order(table: Dict[int, str]) -> List[str]:
    """CLIP's vocabulary order of the 256 base symbols: the self-mapped bytes first (ascending), then the remapped."""
    kept = [b for b in range(256) if table[b] == chr(b)]
    moved = [b for b in range(256) if table[b] != chr(b)]
    return [table[b] for b in kept + moved]


class BpeTokenizer:
    def __init__(self, vocab_path: Optional[str] = None, context_length: int = 77):
        vocab_path = vocab_path or os.environ.get("SC_BPE_VOCAB")
        if not vocab_path or not os.path.isfile(vocab_path):
            raise FileNotFoundError(
                "BPE merge table not found: pass vocab_path= or set SC_BPE_VOCAB to OpenAI CLIP's "
                "bpe_simple_vocab_16e6.txt.gz (it ships with open_clip: src/open_clip/bpe_simple_vocab_16e6.txt.gz)")
        opener = gzip.open if vocab_path.endswith(".gz") else open
        with opener(vocab_path, "rb") as f:
            lines = f.read().decode("utf-8").split("\n")
        merges: List[Tuple[str, str]] = []
        for line in lines[1:1 + N_MERGES]:                      # line 0 is a version header
            parts = line.split()
            if len(parts) == 2:
                merges.append((parts[0], parts[1]))
        self.byte_sym = _byte_symbols()
        self.sym_byte = {v: k for k, v in self.byte_sym.items()}
        base = _vocab_order(self.byte_sym)
        vocab = base + [s + "</w>" for s in base] + [a + b for a, b in merges] + [SOT, EOT]
        self.encoder: Dict[str, int] = {tok: i for i, tok in enumerate(vocab)}
        self.decoder: Dict[int, str] = {i: tok for tok, i in self.encoder.items()}
        self.rank: Dict[Tuple[str, str], int] = {m: i for i, m in enumerate(merges)}
        self.vocab_size = len(vocab)
        self.sot_token_id, self.eot_token_id = self.encoder[SOT], self.encoder[EOT]
        self.context_length = context_length
        self._split = regex.compile(
            r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
        self._cache: Dict[str, List[int]] = {}
        try:
            import ftfy  # noqa: F401
            self._fix = ftfy.fix_text
        except Exception:
            self._fix = None

    # ------------------------------------------------------------------ text -> ids
    def clean(self, text: str) -> str:
        if self._fix is not None:
            text = self._fix(text)
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split()).strip().lower()

    def _merge_word(self, symbols: List[str]) -> List[str]:
        """Greedy BPE: repeatedly fuse every occurrence of the adjacent pair with the lowest merge rank."""
        while len(symbols) > 1:
            best, best_rank = None, None
            for pair in zip(symbols, symbols[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            out, i = [], 0
            while i < len(symbols):
                if i + 1 < len(symbols) and symbols[i] == best[0] and symbols[i + 1] == best[1]:
                    out.append(best[0] + best[1])
                    i += 2
                else:
                    out.append(symbols[i])
                    i += 1
            symbols = out
        return symbols

    def _word_ids(self, word: str) -> List[int]:
        ids = self._cache.get(word)
        if ids is None:
            if word in (SOT, EOT):
                ids = [self.encoder[word]]
            else:
                syms = [self.byte_sym[b] for b in word.encode("utf-8")]
                syms[-1] = syms[-1] + "</w>"                       # end-of-word marker on the last symbol
                ids = [self.encoder[s] for s in self._merge_word(syms)]
            self._cache[word] = ids
        return ids

    def encode(self, text: str) -> List[int]:
        out: List[int] = []
        for word in self._split.findall(self.clean(text)):
            out.extend(self._word_ids(word))
        return out

    def decode(self, ids: Iterable[int]) -> str:
        text = "".join(self.decoder[int(i)] for i in ids)
        data = bytearray(self.sym_byte.get(c, 32) for c in text.replace("</w>", " "))
        return data.decode("utf-8", errors="replace")

    def __call__(self, texts: Union[str, Sequence[str]], context_length: Optional[int] = None) -> torch.Tensor:
        """int64 [n, context_length]: SOT, tokens, EOT, zero padding; over-long inputs are cut and end with EOT."""
        if isinstance(texts, str):
            texts = [texts]
        n_ctx = context_length or self.context_length
        out = torch.zeros((len(texts), n_ctx), dtype=torch.int64)
        for row, text in enumerate(texts):
            ids = [self.sot_token_id] + self.encode(text) + [self.eot_token_id]
            if len(ids) > n_ctx:
                ids = ids[:n_ctx]
                ids[-1] = self.eot_token_id
            out[row, :len(ids)] = torch.tensor(ids, dtype=torch.int64)
        return out
