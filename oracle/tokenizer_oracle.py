"""ORACLE — test infrastructure, not product code.

Pure-Python restatement of BERT/MPNet WordPiece tokenisation as sentence-transformers applies it before
`encode`: BertNormalizer(clean_text, handle_chinese_chars, lowercase + strip accents) ->
BertPreTokenizer (whitespace + punctuation split) -> greedy longest-match WordPiece ('##' continuation,
max 100 chars per word -> unk) -> specials -> truncation to max_len (keep the first max_len-2 pieces).
Follows transformers 5.15.0 models/bert/tokenization_bert_legacy.py (BasicTokenizer :~290-430,
WordpieceTokenizer :~440-490) and models/mpnet/tokenization_mpnet.py:108-163 for the MPNet specials.
"parity unpinned" against the reference (no tokenizer fixtures there); pinned against the HF `tokenizers`
pipeline in tests/test_host_cli.py.  Only tests/ may import this.
"""
from __future__ import annotations

import unicodedata
from typing import Dict, List


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def basic_tokenize(text: str, lowercase: bool = True) -> List[str]:
    out = []
    for ch in text:                                   # clean_text
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD or _is_control(ch):
            continue
        out.append(" " if _is_whitespace(ch) else ch)
    text = "".join(out)
    text = "".join(f" {c} " if _is_cjk(ord(c)) else c for c in text)
    words = []
    for tok in text.split():
        if lowercase:
            tok = tok.lower()
            tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
        cur = ""
        for c in tok:
            if _is_punct(c):
                if cur:
                    words.append(cur); cur = ""
                words.append(c)
            else:
                cur += c
        if cur:
            words.append(cur)
    return words


def wordpiece(word: str, vocab: Dict[str, int], unk: str, max_chars: int = 100) -> List[str]:
    if len(word) > max_chars:
        return [unk]
    pieces, start = [], 0
    while start < len(word):
        end, cur = len(word), None
        while start < end:
            sub = word[start:end]
            if start > 0:
                sub = "##" + sub
            if sub in vocab:
                cur = sub
                break
            end -= 1
        if cur is None:
            return [unk]
        pieces.append(cur)
        start = end
    return pieces


def encode(text: str, vocab: Dict[str, int], arch_mpnet: bool, max_len: int, lowercase: bool = True) -> List[int]:
    bos, eos, unk = ("<s>", "</s>", "<unk>" if "<unk>" in vocab else "[UNK]") if arch_mpnet else ("[CLS]", "[SEP]", "[UNK]")
    ids = []
    for w in basic_tokenize(text, lowercase):
        ids.extend(vocab[p] for p in wordpiece(w, vocab, unk))
    ids = ids[:max(0, max_len - 2)]
    return [vocab[bos]] + ids + [vocab[eos]]
