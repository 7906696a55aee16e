#!/usr/bin/env python3
"""Wall time of `HipSentenceEncoder.encode([one query text], low_latency=True)` end to end (tokenizer, batch assembly, H2D, forward, D2H),
next to the forward alone — how much of a single query's latency is host code.  python tools/query_e2e_latency.py"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arxiv_rag_amd import config as C
from arxiv_rag_amd.encoder import HipSentenceEncoder
from arxiv_rag_amd.tokenizer import WordPieceTokenizer
from arxiv_rag_amd.weights import seeded_state_dict
from tests.helpers import synthetic_vocab

cfg = C.MPNET_BASE
small = C.EncoderConfig(**{**cfg.__dict__, "vocab_size": 2000})
vocab = synthetic_vocab(small)
toks = sorted(vocab, key=vocab.get) + [f"[unused{i}]" for i in range(cfg.vocab_size - len(vocab))]
tok = WordPieceTokenizer.from_vocab({t: i for i, t in enumerate(toks)}, cfg)
model = HipSentenceEncoder(cfg, seeded_state_dict(cfg, seed=0), tok)
words = [w for w in vocab if w.isalpha() and len(w) > 1][:200]
rs = np.random.RandomState(0)
out = {}
for nq in (1, 8):
    qs = [" ".join(rs.choice(words, size=9)) for _ in range(nq)]
    for ll in (False, True):
        for _ in range(5): model.encode(qs, normalize_embeddings=True, low_latency=ll)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter(); e = model.encode(qs, normalize_embeddings=True, low_latency=ll); ts.append(time.perf_counter() - t0)
        out[f"nq={nq},{'small_batch' if ll else 'default'}_encode_call_ms"] = round(float(np.median(ts)) * 1e3, 3)
    seqs = model.tokenize(qs)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); model.tokenize(qs); ts.append(time.perf_counter() - t0)
    out[f"nq={nq},tokenize_ms"] = round(float(np.median(ts)) * 1e3, 3)
    out[f"nq={nq},tokens"] = int(sum(len(s) for s in seqs))
print(json.dumps(out))
