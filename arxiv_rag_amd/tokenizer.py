"""Host tokenisation for the encode path.

sentence-transformers tokenises with the HF `tokenizers` pipeline of the checkpoint
(BertNormalizer(lowercase, clean_text) -> BertPreTokenizer -> WordPiece('##') -> "<s> … </s>" for MPNet,
"[CLS] … [SEP]" for BERT; TF tokenization_mpnet.py:108-163), truncating to `max_seq_length`.
This wrapper drives the same library from LOCAL files only (`tokenizer.json`, else `vocab.txt`):
nothing is fetched by name.
"""
from __future__ import annotations

import ctypes
import json
import os
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .config import ARCH_MPNET, EncoderConfig


def _blob(strings: Sequence[bytes]) -> Tuple[bytes, np.ndarray]:
    off = np.zeros(len(strings) + 1, np.int64)
    if strings:
        np.cumsum(np.fromiter(map(len, strings), np.int64, len(strings)), out=off[1:])
    return b"".join(strings), off


class _NativeWordPiece:
    """arx_wp_* (include/arx.h, csrc/wordpiece.cpp): the ASCII restriction of the BERT WordPiece pipeline, multi-threaded,
    writing an id matrix directly.  Built only when the HF pipeline is exactly the one it restates (`_native_spec`)."""

    def __init__(self, vocab: dict, unk: int, bos: int, eos: int, pad: int, lowercase: bool, max_chars: int, triggers: Sequence[str]):
        from . import _lib
        self._libmod = _lib
        self.lib = _lib.load()
        toks = [b""] * (max(vocab.values()) + 1)
        for t, i in vocab.items():
            toks[i] = t.encode("utf-8")
        vb, vo = _blob(toks)
        tb, to = _blob([t.encode("utf-8") for t in triggers])
        self._h = ctypes.c_void_p(None)
        rc = self.lib.arx_wp_create(vb, vo.ctypes.data, len(toks), unk, bos, eos, pad, int(lowercase), max_chars,
                                    tb, to.ctypes.data, len(triggers), ctypes.byref(self._h))
        if rc != 0:
            raise _lib.ArxError(f"arx_wp_create failed (rc={rc})")
        self.pad = pad
        self.threads = max(1, min(32, (os.cpu_count() or 8)))

    MAX_CACHED_SEGMENTS = 4_000_000

    def _encode_raw(self, enc: Sequence[bytes], max_len: int):
        n = len(enc)
        blob, off = _blob(enc)
        ids = np.empty((n, max_len), np.int32)
        lens = np.zeros(n, np.int32)
        fb = np.zeros(n, np.uint8)
        rc = self.lib.arx_wp_encode(self._h, blob, off.ctypes.data, n, max_len, ids.ctypes.data, lens.ctypes.data,
                                    fb.ctypes.data, self.threads)
        if rc != 0:
            raise self._libmod.ArxError(f"arx_wp_encode failed (rc={rc})")
        return ids, lens, fb

    def misses(self) -> List[bytes]:
        ns, nb = ctypes.c_int64(0), ctypes.c_int64(0)
        self.lib.arx_wp_miss_count(self._h, ctypes.byref(ns), ctypes.byref(nb))
        if ns.value == 0:
            return []
        buf = ctypes.create_string_buffer(max(1, nb.value))
        off = np.zeros(ns.value + 1, np.int64)
        self.lib.arx_wp_miss_fetch(self._h, buf, off.ctypes.data)
        raw = buf.raw
        return [raw[off[i]:off[i + 1]] for i in range(ns.value)]

    def teach(self, segments: Sequence[bytes], pieces: Sequence[Sequence[int]]):
        sb, so = _blob(list(segments))
        io = np.zeros(len(pieces) + 1, np.int64)
        if pieces:
            np.cumsum(np.fromiter(map(len, pieces), np.int64, len(pieces)), out=io[1:])
        flat = np.fromiter((t for p in pieces for t in p), np.int32, int(io[-1])) if io[-1] else np.zeros(1, np.int32)
        rc = self.lib.arx_wp_cache_add(self._h, sb, so.ctypes.data, len(segments), flat.ctypes.data, io.ctypes.data)
        if rc != 0:
            raise self._libmod.ArxError(f"arx_wp_cache_add failed (rc={rc})")

    def encode(self, texts: Sequence[str], max_len: int, resolve=None):
        """-> (ids, lens, fallback).  `resolve(list of segment strings) -> list of piece-id lists` is the reference pipeline applied
        to lone segments (no specials, no truncation); with it, texts whose only obstacle was an unseen non-ASCII segment are
        re-encoded after the segments have been learnt, and come back with fallback 0."""
        enc = [t.encode("utf-8") for t in texts]
        ids, lens, fb = self._encode_raw(enc, max_len)
        todo = np.flatnonzero(fb == 2)
        if todo.size and resolve is not None and self.lib.arx_wp_cache_size(self._h) < self.MAX_CACHED_SEGMENTS:
            segs = self.misses()
            self.teach(segs, resolve([s.decode("utf-8") for s in segs]))
            ids2, lens2, fb2 = self._encode_raw([enc[i] for i in todo], max_len)
            ids[todo], lens[todo], fb[todo] = ids2, lens2, fb2
        fb[fb == 2] = 1                                          # anything still unresolved takes the whole-text fallback
        return ids, lens, fb

    def __del__(self):
        try:
            if self._h:
                self.lib.arx_wp_destroy(self._h)
        except Exception:
            pass


def _native_spec(tok) -> Optional[dict]:
    """The native tokenizer is used only when the HF pipeline is exactly BertNormalizer(clean_text) -> BertPreTokenizer ->
    WordPiece('##') -> TemplateProcessing '<bos> $A <eos>' (one special on each side); anything else keeps the HF path."""
    try:
        j = json.loads(tok.to_str())
        nm, pt, md, pp = j.get("normalizer") or {}, j.get("pre_tokenizer") or {}, j.get("model") or {}, j.get("post_processor") or {}
        if nm.get("type") != "BertNormalizer" or not nm.get("clean_text", False) or not nm.get("handle_chinese_chars", True):
            return None
        lowercase = bool(nm.get("lowercase", True))
        if nm.get("strip_accents") not in (None, lowercase):      # accents only matter off-ASCII, but keep the pipelines identical
            return None
        if pt.get("type") != "BertPreTokenizer" or md.get("type") != "WordPiece" or md.get("continuing_subword_prefix") != "##":
            return None
        if pp.get("type") != "TemplateProcessing":
            return None
        single = pp.get("single") or []
        if len(single) != 3 or "SpecialToken" not in single[0] or "Sequence" not in single[1] or "SpecialToken" not in single[2]:
            return None
        sp = pp.get("special_tokens") or {}
        bos = sp[single[0]["SpecialToken"]["id"]]["ids"][0]
        eos = sp[single[2]["SpecialToken"]["id"]]["ids"][0]
        vocab = md["vocab"]
        unk = vocab[md["unk_token"]]
        triggers = [a["content"] for a in j.get("added_tokens") or []]
        pad = j.get("padding") or {}
        return {"vocab": vocab, "unk": unk, "bos": bos, "eos": eos, "lowercase": lowercase,
                "max_chars": int(md.get("max_input_chars_per_word", 100)), "triggers": triggers, "pad_hint": pad.get("pad_id")}
    except Exception:
        return None


class WordPieceTokenizer:
    def __init__(self, tok, cfg: EncoderConfig):
        self._tok = tok
        self.cfg = cfg
        self._native = None
        self.use_native = os.environ.get("ARX_NATIVE_TOKENIZER", "1") != "0"
        spec = _native_spec(tok) if self.use_native else None
        if spec is not None and spec["max_chars"] <= 100:
            self._native = _NativeWordPiece(spec["vocab"], spec["unk"], spec["bos"], spec["eos"], cfg.pad_id, spec["lowercase"],
                                            spec["max_chars"], spec["triggers"])

    @classmethod
    def from_dir(cls, path: str | Path, cfg: EncoderConfig) -> "WordPieceTokenizer":
        from tokenizers import Tokenizer
        p = Path(path)
        if (p / "tokenizer.json").exists():
            tok = Tokenizer.from_file(str(p / "tokenizer.json"))
            tok.no_padding()
            return cls(tok, cfg)
        if (p / "vocab.txt").exists():
            vocab = {}
            for i, line in enumerate((p / "vocab.txt").read_text(encoding="utf-8").split("\n")):
                w = line.rstrip("\r")
                if w != "" or i == 0:
                    vocab.setdefault(w, i)
            return cls.from_vocab(vocab, cfg)
        raise FileNotFoundError(f"{p}: neither tokenizer.json nor vocab.txt found (tokenizers are never fetched by name)")

    @classmethod
    def from_vocab(cls, vocab: dict, cfg: EncoderConfig, lowercase: bool = True) -> "WordPieceTokenizer":
        from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
        if cfg.arch == ARCH_MPNET:
            bos, eos, unk = "<s>", "</s>", "<unk>" if "<unk>" in vocab else "[UNK]"
        else:
            bos, eos, unk = "[CLS]", "[SEP]", "[UNK]"
        for t in (bos, eos, unk):
            if t not in vocab:
                raise ValueError(f"vocab lacks special token {t!r}")
        tok = Tokenizer(models.WordPiece(vocab=vocab, unk_token=unk, max_input_chars_per_word=100))
        tok.normalizer = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None,
                                                    lowercase=lowercase)
        tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
        tok.post_processor = processors.TemplateProcessing(
            single=f"{bos} $A {eos}", pair=f"{bos} $A {eos} {eos} $B {eos}",
            special_tokens=[(bos, vocab[bos]), (eos, vocab[eos])])
        return cls(tok, cfg)

    def encode_batch(self, texts: Sequence[str], max_len: int) -> List[List[int]]:
        """-> token-id lists incl. the two specials, truncated (longest_first == keep the first
        max_len-2 word pieces for single sentences)."""
        self._tok.enable_truncation(max_length=max_len)
        return [e.ids for e in self._tok.encode_batch(list(texts))]

    def _segment_pieces(self, segments: Sequence[str]) -> List[List[int]]:
        """The HF pipeline on lone whitespace-delimited segments: no specials, no truncation (what the native cache stores)."""
        self._tok.no_truncation()
        return [e.ids for e in self._tok.encode_batch(list(segments), add_special_tokens=False)]

    def encode_batch_packed(self, texts: Sequence[str], max_len: int) -> Tuple[np.ndarray, np.ndarray]:
        """-> (ids int32 [n, max_len] right-padded with cfg.pad_id, lens int32 [n]); same ids as `encode_batch`.
        Pure-ASCII texts go through the native multi-threaded tokenizer, the rest (and everything, when the HF pipeline is not the
        plain BERT WordPiece one) through HF `tokenizers`."""
        texts = list(texts)
        if self._native is None:
            seqs = self.encode_batch(texts, max_len)
            ids = np.full((len(seqs), max_len), self.cfg.pad_id, np.int32)
            lens = np.fromiter(map(len, seqs), np.int32, len(seqs))
            for i, s in enumerate(seqs):
                ids[i, :len(s)] = s
            return ids, lens
        ids, lens, fb = self._native.encode(texts, max_len, resolve=self._segment_pieces)
        rest = np.flatnonzero(fb)
        if rest.size:
            seqs = self.encode_batch([texts[i] for i in rest], max_len)
            for i, s in zip(rest, seqs):
                ids[i, :len(s)] = s
                ids[i, len(s):] = self.cfg.pad_id
                lens[i] = len(s)
        return ids, lens


class TokenIdPassthrough:
    """For synthetic runs: 'texts' are already whitespace-separated token ids."""

    def __init__(self, cfg: EncoderConfig):
        self.cfg = cfg

    def encode_batch(self, texts, max_len):
        return [[int(t) for t in s.split()][:max_len] for s in texts]
