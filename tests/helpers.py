"""Test-only helpers: an oracle-backed stand-in for the encoder so that the host logic (CLI, loader,
writer, order contract, fallbacks, multi-rank sharding) can be exercised on CPU.  Never imported by
the product package."""
import json
import numpy as np

from arxiv_rag_amd import config as C
from arxiv_rag_amd.weights import seeded_state_dict
from oracle import encoder_oracle as EO


def synthetic_vocab(cfg, n_words=90):
    """specials + single characters + seeded pseudo-words/continuations; len == cfg.vocab_size."""
    rs = np.random.RandomState(7)
    if cfg.arch == C.ARCH_MPNET:
        toks = ["<s>", "<pad>", "</s>", "<unk>"]
    else:
        toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]"]
    letters = "abcdefghijklmnopqrstuvwxyz"
    toks += list(letters) + ["##" + c for c in letters] + list(".,;!?-()")
    seen = set(toks)
    while len(toks) < cfg.vocab_size:
        w = "".join(rs.choice(list(letters), size=rs.randint(2, 6)))
        w = w if rs.rand() < 0.6 else "##" + w
        if w not in seen:
            seen.add(w); toks.append(w)
    return {t: i for i, t in enumerate(toks)}


class OracleSentenceModel:
    """Same two methods as SentenceTransformer / HipSentenceEncoder, arithmetic by the numpy oracle."""

    def __init__(self, cfg, sd, tokenizer, fail_on=None):
        self.cfg, self.sd, self.tokenizer, self.fail_on = cfg, sd, tokenizer, fail_on or (lambda texts: False)
        self.calls = []

    def get_sentence_embedding_dimension(self):
        return self.cfg.hidden

    def encode(self, sentences, batch_size=32, normalize_embeddings=False, **kw):
        self.calls.append(len(sentences))
        if self.fail_on(sentences):
            raise RuntimeError("injected failure")
        seqs = self.tokenizer.encode_batch(list(sentences), self.cfg.max_seq_length)
        return EO.encode_ragged(self.sd, self.cfg, seqs, batch_size=batch_size, normalize=normalize_embeddings)


def make_chunk_tree(root, n_files=12, chunks_per_file=5, seed=0, words=None):
    """Synthetic stage-3 output: chunk JSON files with the schema of pipeline.py:369-387 + quality_score."""
    rs = np.random.RandomState(seed)
    words = words or ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa"]
    root.mkdir(parents=True, exist_ok=True)
    all_chunks = []
    for f in range(n_files):
        pid = f"0704.{f:04d}"
        chunks = []
        for c in range(chunks_per_file):
            text = " ".join(rs.choice(words, size=rs.randint(3, 40)))
            chunks.append({"chunk_id": f"{pid}_chunk_{c}", "text": text,
                           "metadata": {"quality_score": float(np.round(rs.uniform(0.8, 1.0), 3)), "paper_id": pid,
                                        "section": rs.choice(["Introduction", "Methods", "Results"]).item(), "chunk_index": c}})
        sub = root / ("a" if f % 2 else "b")
        sub.mkdir(exist_ok=True)
        (sub / f"{pid}.json").write_text(json.dumps({"paper_id": pid, "chunks": chunks}))
        all_chunks.extend(chunks)
    return all_chunks
