"""Host tokenisation for the encode path.

sentence-transformers tokenises with the HF `tokenizers` pipeline of the checkpoint
(BertNormalizer(lowercase, clean_text) -> BertPreTokenizer -> WordPiece('##') -> "<s> … </s>" for MPNet,
"[CLS] … [SEP]" for BERT; TF tokenization_mpnet.py:108-163), truncating to `max_seq_length`.
This wrapper drives the same library from LOCAL files only (`tokenizer.json`, else `vocab.txt`):
nothing is fetched by name.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Sequence

from .config import ARCH_MPNET, EncoderConfig


class WordPieceTokenizer:
    def __init__(self, tok, cfg: EncoderConfig):
        self._tok = tok
        self.cfg = cfg

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


class TokenIdPassthrough:
    """For synthetic runs: 'texts' are already whitespace-separated token ids."""

    def __init__(self, cfg: EncoderConfig):
        self.cfg = cfg

    def encode_batch(self, texts, max_len):
        return [[int(t) for t in s.split()][:max_len] for s in texts]
