#!/usr/bin/env python3
"""End-to-end run of the drop-in script on the GPU box: synthetic stage-3 chunk tree (JSON files) -> loader -> HIP encoder
(mpnet-base shape, seeded weights, synthetic vocabulary) -> embeddings_saved/{embeddings.npy float64, metadata.json, index.json}.
Prints the stage times the script itself reports plus chunks/s over the whole run.  usage: cli_bench.py [n_files] [chunks_per_file]"""
import contextlib, io, json, os, re, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import numpy as np
from arxiv_rag_amd import config as C, generate_embeddings_parallel as GEN
from arxiv_rag_amd.encoder import HipSentenceEncoder
from arxiv_rag_amd.tokenizer import WordPieceTokenizer
from arxiv_rag_amd.weights import seeded_state_dict
from tests.helpers import synthetic_vocab

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cpf = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = C.MPNET_BASE
vocab = synthetic_vocab(C.EncoderConfig(**{**cfg.__dict__, "vocab_size": 8000}))
toks = sorted(vocab, key=vocab.get); toks += [f"[unused{i}]" for i in range(cfg.vocab_size - len(toks))]
vocab = {t: i for i, t in enumerate(toks)}
words = [w for w in vocab if w.isalpha() and len(w) > 1][:3000]
rs = np.random.RandomState(0)
with tempfile.TemporaryDirectory() as td:
    td = Path(td); tree = td / "chunks"; tree.mkdir()
    t0 = time.time()
    for f in range(n_files):
        pid = f"0704.{f:05d}"
        chunks = [{"chunk_id": f"{pid}_chunk_{c}", "text": " ".join(rs.choice(words, size=rs.randint(60, 260))),
                   "metadata": {"quality_score": 0.95, "paper_id": pid, "section": "Methods", "chunk_index": c}} for c in range(cpf)]
        (tree / f"{pid}.json").write_text(json.dumps({"paper_id": pid, "chunks": chunks}))
    print(f"tree: {n_files} files x {cpf} chunks written in {time.time() - t0:.1f} s")
    model = HipSentenceEncoder(cfg, seeded_state_dict(cfg, seed=0), WordPieceTokenizer.from_vocab(vocab, cfg))
    model.encode(["warm up"] * 2048, batch_size=200, normalize_embeddings=True)
    os.chdir(td)
    buf = io.StringIO()
    t0 = time.time()
    argv = [str(tree), "--min-quality", "0.9", "--skip-chroma"]
    if os.environ.get("CLI_QUERIES"):             # also run the added search step over the shard the encode step leaves in HBM
        (td / "queries.txt").write_text("\n".join(" ".join(rs.choice(words, size=12)) for _ in range(int(os.environ["CLI_QUERIES"]))) + "\n")
        argv += ["--queries", str(td / "queries.txt")]
    with contextlib.redirect_stdout(buf):
        rc = GEN.main(argv, model_factory=lambda name: model)
    dt = time.time() - t0
    out = buf.getvalue()
    n = n_files * cpf
    stages = dict(re.findall(r"(Loading|Embedding generation) completed in ([\d.]+) seconds", out))
    arr = np.load(td / "embeddings_saved" / "embeddings.npy", mmap_mode="r")
    print(json.dumps({"rc": rc, "chunks": n, "total_s": round(dt, 2), "chunks_per_s_whole_script": round(n / dt, 1),
                      "load_s": float(stages.get("Loading", 0)), "embed_s": float(stages.get("Embedding generation", 0)),
                      "write_s": round(dt - float(stages.get("Loading", 0)) - float(stages.get("Embedding generation", 0)), 2),
                      "npy": [str(arr.dtype), list(arr.shape)], "search_step": bool(os.environ.get("CLI_QUERIES"))}))
