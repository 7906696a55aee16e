#!/usr/bin/env python3
"""End-to-end text -> embeddings rate of the drop-in path on the GPU box (host tokenisation + H2D + encode + D2H),
next to bench.py's device-resident number.  Synthetic chunk texts, synthetic WordPiece vocab, mpnet-base shape."""
import sys, time, json, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from arxiv_rag_amd import config as C
from arxiv_rag_amd.weights import seeded_state_dict
from arxiv_rag_amd.tokenizer import WordPieceTokenizer
from arxiv_rag_amd.encoder import HipSentenceEncoder
from tests.helpers import synthetic_vocab

cfg = C.MPNET_BASE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
vocab = synthetic_vocab(C.EncoderConfig(**{**cfg.__dict__, "vocab_size": 8000}))
toks = sorted(vocab, key=vocab.get); toks += [f"[unused{i}]" for i in range(cfg.vocab_size - len(toks))]
vocab = {t: i for i, t in enumerate(toks)}
tok = WordPieceTokenizer.from_vocab(vocab, cfg)
words = [w for w in vocab if w.isalpha() and len(w) > 1][:3000]
rs = np.random.RandomState(0)
texts = [" ".join(rs.choice(words, size=rs.randint(60, 260))) for _ in range(n)]
model = HipSentenceEncoder(cfg, seeded_state_dict(cfg, seed=0), tok)
model.encode(texts[:2048], batch_size=1024, normalize_embeddings=True)           # warm
t0 = time.time(); seqs = model.tokenize(texts); t_tok_hf = time.time() - t0
t0 = time.time(); _, lens_ = tok.encode_batch_packed(texts, cfg.max_seq_length); t_tok = time.time() - t0
ntok = sum(len(s) for s in seqs)
assert int(lens_.sum()) == ntok
print(json.dumps({"tokenizer_hf_s": round(t_tok_hf, 3), "tokenizer_native_s": round(t_tok, 3), "native": tok._native is not None}))
for bs in (200, 1024):
    t0 = time.time(); emb = model.encode(texts, batch_size=bs, normalize_embeddings=True); t = time.time() - t0
    print(json.dumps({"chunks": n, "tokens": ntok, "mean_tokens": round(ntok / n, 1), "batch_size": bs,
                      "tokenize_s": round(t_tok, 3), "tokens_per_s_tokenizer": round(ntok / t_tok),
                      "encode_total_s": round(t, 3), "chunks_per_s_end_to_end": round(n / t, 1)}))
