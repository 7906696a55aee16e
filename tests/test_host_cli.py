"""Host logic on CPU: loader / writer against the reference's own functions' output (tests/golden/harness,
made by tools/make_golden.py from GEN:76-92 and GEN:271-321), CLI flags/exit codes/order/fallbacks with an
oracle-backed model injected, tokenizer vs the pure-Python restatement, C-ABI exports, world-size-2 gloo."""
import ctypes
import hashlib
import json
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from arxiv_rag_amd import config as C
from arxiv_rag_amd import generate_embeddings_parallel as GEN
from arxiv_rag_amd.tokenizer import WordPieceTokenizer
from arxiv_rag_amd.weights import seeded_state_dict
from oracle import encoder_oracle as EO, tokenizer_oracle as TO
from tests.helpers import OracleSentenceModel, make_chunk_tree, synthetic_vocab

ROOT = Path(__file__).resolve().parents[1]
HARN = ROOT / "tests" / "golden" / "harness"


# ------------------------------------------------------------------ loader / writer vs the reference's own output
def test_loader_matches_reference_output():
    files = sorted(f for f in (HARN / "input").rglob("*.json") if not f.name.startswith("._"))
    got = []
    for f in files:
        got.extend(GEN.load_chunks_from_file(f, min_quality=0.9))
    assert [c.get("chunk_id") for c in got] == json.loads((HARN / "expected_loaded_ids.json").read_text())
    allc = GEN.load_chunks_parallel(HARN / "input", min_quality=0.9, num_workers=2)
    assert [c.get("chunk_id") for c in allc] == [c.get("chunk_id") for c in got]       # ._ file skipped, broken file ignored


def test_writer_bytes_match_reference_output(tmp_path):
    files = sorted(f for f in (HARN / "input").rglob("*.json") if not f.name.startswith("._"))
    chunks = [c for f in files for c in GEN.load_chunks_from_file(f, 0.9)]
    embs = list(np.load(HARN / "input_embeddings_f32.npy"))
    embs[1] = np.zeros(8)                                   # float64 zero-vector fallback row (GEN:169)
    GEN.save_embeddings_to_disk_fallback(chunks, embs, output_dir=str(tmp_path))
    for n in ("metadata.json", "index.json", "embeddings.npy"):
        assert (tmp_path / n).read_bytes() == (HARN / f"expected_{n}").read_bytes(), n
    meta = json.loads((HARN / "expected_embeddings_meta.json").read_text())
    arr = np.load(tmp_path / "embeddings.npy")
    assert str(arr.dtype) == meta["dtype"] == "float64" and list(arr.shape) == meta["shape"]
    assert hashlib.sha256(arr.tobytes()).hexdigest() == meta["sha256"]


def test_cli_flags_and_defaults_match_reference():
    """flag names and defaults of GEN:471-491 (read from the reference text when it is present)."""
    p = GEN.build_parser()
    d = vars(p.parse_args(["some_dir"]))
    want = {"model": "all-mpnet-base-v2", "min_quality": 0.9, "batch_size": 200, "chroma_db": "./chroma_db",
            "collection_name": "scientific_papers", "load_workers": None, "embedding_workers": None,
            "store_batch_size": 2000, "chunks_per_worker": 500}
    for k, v in want.items():
        assert d[k] == v, k
    ref = Path("/root/reference/4-embed/generation/generate_embeddings_parallel.py")
    if ref.exists():
        flags = set(re.findall(r"add_argument\('(--[a-z-]+)'", ref.read_text()))
        ours = {a for act in p._actions for a in act.option_strings}
        assert flags <= ours, flags - ours


# ------------------------------------------------------------------ CLI end to end (cfg-1 style plumbing, scaled to CPU)
@pytest.fixture()
def tiny_model():
    cfg = C.TINY_BERT
    sd = seeded_state_dict(cfg, seed=4, std=0.05)
    tok = WordPieceTokenizer.from_vocab(synthetic_vocab(cfg), cfg)
    return cfg, sd, tok


def test_cli_end_to_end_order_layout_and_filter(tmp_path, tiny_model, monkeypatch):
    cfg, sd, tok = tiny_model
    chunks = make_chunk_tree(tmp_path / "in", n_files=10, chunks_per_file=6)
    monkeypatch.chdir(tmp_path)
    model = OracleSentenceModel(cfg, sd, tok)
    rc = GEN.main([str(tmp_path / "in"), "--model", "tiny-bert", "--batch-size", "7", "--chunks-per-worker", "16",
                   "--min-quality", "0.9", "--skip-chroma"], model_factory=lambda name: model)
    assert rc == 0
    kept = GEN.load_chunks_parallel(tmp_path / "in", 0.9, 2)
    assert 0 < len(kept) < len(chunks) and all(c["metadata"]["quality_score"] >= 0.9 for c in kept)
    arr = np.load(tmp_path / "embeddings_saved" / "embeddings.npy")
    assert arr.dtype == np.float64 and arr.shape == (len(kept), cfg.hidden) and not np.isfortran(arr)
    meta = json.loads((tmp_path / "embeddings_saved" / "metadata.json").read_text())
    assert [m["chunk_id"] for m in meta] == [c["chunk_id"] for c in kept]
    assert list(meta[0].keys()) == ["chunk_id", "paper_id", "section", "quality_score", "text", "text_length"]
    idx = json.loads((tmp_path / "embeddings_saved" / "index.json").read_text())
    assert idx == {"total_embeddings": len(kept), "embedding_dimension": cfg.hidden, "total_size_gb": arr.nbytes / 2**30}
    # row i <-> chunk i, independent of the quantum / sub-batch split
    ref = EO.encode_ragged(sd, cfg, tok.encode_batch([c["text"] for c in kept], cfg.max_seq_length), batch_size=64)
    cos = (arr * ref).sum(1) / (np.linalg.norm(arr, axis=1) * np.linalg.norm(ref, axis=1))
    assert cos.min() > 1 - 1e-6
    assert max(model.calls) <= 7                           # sub-batches honour --batch-size


def test_metadata_prefetch_is_byte_identical_and_leaves_nothing_behind(tmp_path, tiny_model, monkeypatch):
    """metadata.json is serialised in a helper thread while the embeddings are generated and renamed into place by the writer:
    same bytes as the writer's own serialisation, no `.partial` left after a successful run nor after a run that fails first."""
    cfg, sd, tok = tiny_model
    model = OracleSentenceModel(cfg, sd, tok)
    chunks = make_chunk_tree(tmp_path / "in", n_files=4, chunks_per_file=6, seed=8)
    monkeypatch.chdir(tmp_path)
    assert GEN.main([str(tmp_path / "in"), "--min-quality", "0.0", "--skip-chroma"], model_factory=lambda name: model) == 0
    out = tmp_path / "embeddings_saved"
    assert not (out / "metadata.json.partial").exists()
    loaded = GEN.load_chunks_parallel(tmp_path / "in", min_quality=0.0)
    GEN._dump_metadata(loaded, tmp_path / "direct.json")
    assert (out / "metadata.json").read_bytes() == (tmp_path / "direct.json").read_bytes()
    # a prefetch that is never taken (the run died before the writer) removes its partial file
    pf = GEN.MetadataPrefetch(loaded, str(tmp_path / "other"))
    pf.start(); pf.discard()
    assert not (tmp_path / "other" / "metadata.json.partial").exists()
    # a prefetch for a different chunk list is not used
    pf = GEN.MetadataPrefetch(loaded[:3], str(tmp_path / "third")); pf.start()
    GEN.save_embeddings_to_disk_fallback(loaded, [np.zeros(4)] * len(loaded), output_dir=str(tmp_path / "third"), prefetch=pf)
    assert len(json.loads((tmp_path / "third" / "metadata.json").read_text())) == len(loaded)
    assert not (tmp_path / "third" / "metadata.json.partial").exists()


def test_cli_exit_codes(tmp_path, tiny_model, monkeypatch):
    cfg, sd, tok = tiny_model
    monkeypatch.chdir(tmp_path)
    assert GEN.main([str(tmp_path / "missing")]) == 1                                   # GEN:497-499
    (tmp_path / "empty").mkdir()
    assert GEN.main([str(tmp_path / "empty"), "--skip-chroma"], model_factory=lambda n: None) == 1   # GEN:527-529
    make_chunk_tree(tmp_path / "in", n_files=2, chunks_per_file=3)
    def boom(name):
        raise FileNotFoundError("no local model")
    assert GEN.main([str(tmp_path / "in"), "--min-quality", "0.0", "--skip-chroma"], model_factory=boom) == 1
    # chromadb absent -> 1 AFTER the disk backup was written (GEN:553-562)
    model = OracleSentenceModel(cfg, sd, tok)
    rc = GEN.main([str(tmp_path / "in"), "--min-quality", "0.0"], model_factory=lambda n: model)
    try:
        import chromadb  # noqa: F401
        assert rc == 0
    except ImportError:
        assert rc == 1
    assert (tmp_path / "embeddings_saved" / "embeddings.npy").exists()


def test_worker_fallbacks_never_raise(tiny_model):
    """batch failure -> per-item retry -> zero row (GEN:155-169); whole-quantum failure -> error tuple (172-177)."""
    cfg, sd, tok = tiny_model
    texts = ["alpha beta", "POISON gamma", "delta", "epsilon zeta"]
    model = OracleSentenceModel(cfg, sd, tok, fail_on=lambda s: any("POISON" in t for t in s))
    GEN._model, GEN._model_name = model, "m"
    idx, rows, err = GEN.generate_embeddings_worker((texts, "m", 4, 3))
    assert idx == 3 and err is None and len(rows) == 4
    assert np.array_equal(rows[1], np.zeros(cfg.hidden)) and rows[1].dtype == np.float64
    ok = EO.encode_ragged(sd, cfg, tok.encode_batch([texts[0]], cfg.max_seq_length))
    assert np.allclose(rows[0], ok[0], atol=1e-6)
    GEN._model = None; GEN._model_name = None
    def boom(name):
        raise RuntimeError("cannot load")
    import arxiv_rag_amd.generate_embeddings_parallel as G2
    orig = G2.get_worker_model
    G2.get_worker_model = boom
    try:
        idx, rows, err = G2.generate_embeddings_worker((texts, "m", 4, 9))
    finally:
        G2.get_worker_model = orig
    assert idx == 9 and rows == [] and "cannot load" in err


def test_missing_quantum_is_padded_with_zero_rows(tiny_model, monkeypatch):
    cfg, sd, tok = tiny_model
    chunks = [{"text": f"alpha {i}", "metadata": {}} for i in range(10)]
    model = OracleSentenceModel(cfg, sd, tok)
    GEN._model, GEN._model_name = model, "m"
    real = GEN.generate_embeddings_worker
    monkeypatch.setattr(GEN, "generate_embeddings_worker", lambda a: (a[3], [], "dead") if a[3] == 1 else real(a))
    embs = GEN.generate_embeddings_parallel(chunks, "m", batch_size=4, chunks_per_worker=4)
    assert len(embs) == 10
    assert all(np.array_equal(e, np.zeros(cfg.hidden)) for e in embs[-4:])               # padded at the end, as GEN:260-265
    GEN._model = None


# ------------------------------------------------------------------ tokenizer
@pytest.mark.parametrize("preset", ["tiny-mpnet", "tiny-bert"])
def test_tokenizer_matches_python_restatement(preset):
    cfg = C.PRESETS[preset]
    vocab = synthetic_vocab(cfg)
    tok = WordPieceTokenizer.from_vocab(vocab, cfg)
    rs = np.random.RandomState(0)
    words = [w for w in vocab if w.isalpha() and len(w) > 1][:40]
    texts = ["", "   ", "Alpha,beta!  (Gamma)", "Über-naïve café\tx y", "x" * 150 + " ok", "日本 ab"]
    for _ in range(60):
        n = rs.randint(1, 90)
        texts.append(" ".join(rs.choice(words + ["zzzzqq", "A.B", "c-d"], size=n)))
    got = tok.encode_batch(texts, 64)
    for t, g in zip(texts, got):
        assert g == TO.encode(t, vocab, cfg.arch == C.ARCH_MPNET, 64), t
        assert len(g) <= 64 and len(g) >= 2


def test_tokenizer_from_dir_vocab_txt(tmp_path):
    cfg = C.TINY_BERT
    vocab = synthetic_vocab(cfg)
    (tmp_path / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n", encoding="utf-8")
    a = WordPieceTokenizer.from_dir(tmp_path, cfg).encode_batch(["alpha beta, gamma"], 32)
    b = WordPieceTokenizer.from_vocab(vocab, cfg).encode_batch(["alpha beta, gamma"], 32)
    assert a == b


def test_tokenizer_from_dir_tokenizer_json(tmp_path):
    """A checkpoint directory normally carries tokenizer.json (preferred over vocab.txt): same ids as the vocab-built
    pipeline, through both the list API and the packed (native feeder) API."""
    cfg = C.TINY_MPNET
    vocab = synthetic_vocab(cfg)
    WordPieceTokenizer.from_vocab(vocab, cfg)._tok.save(str(tmp_path / "tokenizer.json"))
    (tmp_path / "vocab.txt").write_text("not the vocabulary\n")                      # must be ignored when tokenizer.json exists
    texts = ["alpha beta, gamma", "", "Zeta  ETA!", "x" * 150, " ".join(["ab"] * 100)]
    a = WordPieceTokenizer.from_dir(tmp_path, cfg)
    b = WordPieceTokenizer.from_vocab(vocab, cfg)
    assert a.encode_batch(texts, 32) == b.encode_batch(texts, 32)
    ia, la = a.encode_batch_packed(texts, 32)
    assert [ia[i, :la[i]].tolist() for i in range(len(texts))] == b.encode_batch(texts, 32)
    with pytest.raises(FileNotFoundError):
        WordPieceTokenizer.from_dir(tmp_path / "nowhere", cfg)


def test_loader_process_pool_equals_thread_pool(tmp_path, monkeypatch):
    """Corpus-sized trees are parsed by a spawn-context process pool (json.load holds the GIL); the chunk list must be the
    one the in-process path yields — sorted file order, filter applied, unreadable files skipped."""
    make_chunk_tree(tmp_path / "in", n_files=40, chunks_per_file=3, seed=5)
    (tmp_path / "in" / "a" / "broken.json").write_text("{not json")
    (tmp_path / "in" / "a" / "._0704.0001.json").write_text("{}")
    ref = GEN.load_chunks_parallel(tmp_path / "in", 0.9, 4)
    monkeypatch.setattr(GEN, "PROCESS_POOL_MIN_FILES", 1)
    got = GEN.load_chunks_parallel(tmp_path / "in", 0.9, 4)
    assert got == ref and 0 < len(ref) < 120


# ------------------------------------------------------------------ C ABI
@pytest.mark.parametrize("preset", ["all-mpnet-base-v2", "all-MiniLM-L6-v2"])
def test_native_wordpiece_equals_hf_pipeline_and_oracle(preset):
    """arx_wp_* (csrc/wordpiece.cpp) against the HF `tokenizers` pipeline it restates on ASCII, and against the Python
    restatement in oracle/: fuzzed ASCII (punctuation runs, control bytes, over-long words, upper case), truncation at several
    max_len, and non-ASCII / added-token texts that must take the fallback and still come back identical."""
    cfg = C.PRESETS[preset]
    vocab = synthetic_vocab(C.EncoderConfig(**{**cfg.__dict__, "vocab_size": 6000}))
    tok = WordPieceTokenizer.from_vocab(vocab, cfg)
    assert tok._native is not None, "native tokenizer not engaged for the plain BERT WordPiece pipeline"
    rs = np.random.RandomState(3)
    words = [w for w in vocab if w.isalpha()][:1500]
    alphabet = list("abcXYZ019 .,;!?-()[]{}<>#$%^&*_+=~`'\"/\\|@:\t\n\r\x0b\x0c\x00\x1f\x7f")
    texts = []
    for _ in range(1500):
        parts = []
        for _ in range(rs.randint(0, 50)):
            r = rs.rand()
            if r < 0.6:
                parts.append(rs.choice(words))
            elif r < 0.8:
                parts.append("".join(rs.choice(alphabet, size=rs.randint(1, 8))))
            elif r < 0.85:
                parts.append("x" * rs.randint(95, 110))
            else:
                parts.append(rs.choice(words).upper() + rs.choice(words))
        texts.append((" " if rs.rand() < 0.9 else "").join(parts))
    texts += ["", " ", "\t\n", "é accents ünï", "naïve café", "中文 text", "a" * 100, "b" * 101, "<s> literal", "[UNK] x", "##ab ##"]
    arch_mpnet = cfg.arch == C.ARCH_MPNET
    for max_len in (cfg.max_seq_length, 16, 5, 2):
        want = tok.encode_batch(texts, max_len)                              # HF pipeline
        ids, lens = tok.encode_batch_packed(texts, max_len)                  # native + fallback
        assert ids.shape == (len(texts), max_len) and ids.dtype == np.int32
        for i, w in enumerate(want):
            assert list(ids[i, :lens[i]]) == w, (max_len, texts[i][:60])
            assert (ids[i, lens[i]:] == cfg.pad_id).all()
    for t in texts[:300] + texts[-11:]:
        assert TO.encode(t, vocab, arch_mpnet, cfg.max_seq_length) == list(
            (lambda r: r[0][0, :r[1][0]])(tok.encode_batch_packed([t], cfg.max_seq_length)))
    # the flags themselves: ASCII rows are native; an unseen non-ASCII segment falls back unless a resolver teaches it
    _, _, fb = tok._native.encode(["plain ascii", "zz\u00e9qq", ""], 8)
    assert fb.tolist() == [0, 1, 0]
    _, _, fb = tok._native.encode(["plain ascii", "zz\u00e9qq", ""], 8, resolve=tok._segment_pieces)
    assert fb.tolist() == [0, 0, 0]


def test_native_wordpiece_unicode_segments_equal_hf_pipeline():
    """Non-ASCII text: the native tokenizer learns each distinct whitespace-delimited non-ASCII segment from the HF pipeline once
    and must then reproduce HF's ids for whole texts — accents (composed and decomposed), combining marks after a space, Greek
    incl. final sigma, dotted/dotless i, dashes and quotes, math symbols beyond the BMP, CJK (spaced out by the normaliser),
    no-break / thin / ideographic spaces, zero-width and format characters, U+FFFD, NEL and line separator, ligatures."""
    cfg = C.MPNET_BASE
    vocab = synthetic_vocab(C.EncoderConfig(**{**cfg.__dict__, "vocab_size": 6000}))
    pool = [w for w in vocab if w.isalpha() and len(w) > 4]
    for j, e in enumerate(["\u03b1", "\u03b2", "##\u03b1", "e", "naive", "cafe", "\u2013", "\u201c", "\u4e2d", "\u6587", "ss", "i"]):
        if e not in vocab:
            vocab[e] = vocab.pop(pool[j])
    tok = WordPieceTokenizer.from_vocab(vocab, cfg)
    assert tok._native is not None
    cps = [0x3b1, 0x3b2, 0x3b3, 0xe9, 0xe8, 0xfc, 0xf1, 0xdf, 0x131, 0x130, 0x3a3, 0x3c3, 0x3c2, 0x2013, 0x2014, 0x201c, 0x201d,
           0x2018, 0x2019, 0x2026, 0xb7, 0xd7, 0xf7, 0xb0, 0xb1, 0xb2, 0xbd, 0x2192, 0x221e, 0x2211, 0x222b, 0x2202, 0x221a, 0x2264,
           0x2265, 0x2260, 0x2248, 0x1d465, 0x1d538, 0x4e2d, 0x6587, 0x65e5, 0x672c, 0x8a9e, 0xd55c, 0xad6d, 0x627, 0x628, 0xa0,
           0x2009, 0x200b, 0x200d, 0x301, 0x308, 0xfeff, 0xad, 0xfffd, 0x3000, 0x85, 0x2028, 0xfb01, 0xc5, 0x1c4, 0x1c5]
    uni = [chr(c) for c in cps] + ["A\u030a", "o\u0308"]
    rs = np.random.RandomState(1)
    words = [w for w in vocab if w.isalpha() and w.isascii()][:1500]
    texts = []
    for _ in range(1500):
        parts = []
        for _ in range(rs.randint(0, 40)):
            r = rs.rand()
            if r < 0.55:
                parts.append(rs.choice(words))
            elif r < 0.75:
                parts.append(rs.choice(words)[:3] + "".join(rs.choice(uni, size=rs.randint(1, 4))) + rs.choice(words)[:2])
            elif r < 0.85:
                parts.append("".join(rs.choice(uni, size=rs.randint(1, 6))))
            elif r < 0.9:
                parts.append(rs.choice(list(".,;-()[]\t\n\x00\x0b")))
            else:
                parts.append(rs.choice(words).upper())
        texts.append((" " if rs.rand() < 0.9 else "").join(parts))
    texts += ["\u4e2d\u6587" * 300, "\u00e9" * 200 + " x", "a\u0301", "\u0301a", " \u0301 ", "\u0130stanbul ISTANBUL \u0131i",
              "\u03a3\u0391\u03a3 \u03c3\u03b1\u03c2", "x\u00a0y", "\ufb01nal"]
    for max_len in (cfg.max_seq_length, 12, 3):
        want = tok.encode_batch(texts, max_len)
        ids, lens = tok.encode_batch_packed(texts, max_len)
        for i, w in enumerate(want):
            assert list(ids[i, :lens[i]]) == w, (max_len, ascii(texts[i])[:80])
            assert (ids[i, lens[i]:] == cfg.pad_id).all()
    _, _, fb = tok._native.encode(texts, cfg.max_seq_length, resolve=tok._segment_pieces)
    assert int((fb == 1).sum()) == 1 and fb[-9] == 1            # only the 1800-byte unsegmented CJK run takes the whole-text fallback
    assert tok._native.lib.arx_wp_cache_size(tok._native._h) > 1000


def test_native_wordpiece_respects_added_tokens(tmp_path):
    """A tokenizer.json with added special tokens: texts that contain one of them literally are not tokenised natively."""
    cfg = C.TINY_BERT
    vocab = synthetic_vocab(cfg)
    base = WordPieceTokenizer.from_vocab(vocab, cfg)
    base._tok.add_special_tokens(["[SEP]", "[CLS]", "[UNK]"])
    base._tok.save(str(tmp_path / "tokenizer.json"))
    tok = WordPieceTokenizer.from_dir(tmp_path, cfg)
    assert tok._native is not None
    texts = ["ab cd", "ab [SEP] cd", "x [unk] y", "[CLS]"]
    want = tok.encode_batch(texts, 16)
    ids, lens = tok.encode_batch_packed(texts, 16)
    assert [list(ids[i, :lens[i]]) for i in range(len(texts))] == want
    _, _, fb = tok._native.encode(texts, 16)
    assert fb.tolist() == [0, 1, 0, 1]


def test_cabi_exports_every_declared_symbol():
    from arxiv_rag_amd import _lib
    hdr = (ROOT / "include" / "arx.h").read_text()
    declared = set(re.findall(r"\b(arx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert _lib.LIB_PATH.exists(), "libarx_hip.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/arx.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    bound = _lib.load()
    assert bound.arx_version() == 111
    # host-only entry point: MPNet bucket table vs the golden (transformers) values — no GPU involved
    g = np.load(ROOT / "tests" / "golden" / "mpnet_tables.npz")
    got = np.array([bound.arx_mpnet_bucket(int(d), 32, 128) for d in g["delta"]])
    assert np.array_equal(got, g["bucket_of_delta"])
    assert bound.arx_topk_workspace_bytes(10_000_000, 256, 768, 10) > 0
    assert bound.arx_topk_workspace_bytes(10, 1, 768, 99) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from arxiv_rag_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.ArxError):
        _lib.load()


def test_product_never_imports_oracle():
    for f in list((ROOT / "arxiv_rag_amd").glob("*.py")) + [ROOT / "arxiv_rag_amd" / "csrc" / "build.sh"]:
        assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", f.read_text(), re.M), f


# ------------------------------------------------------------------ world size 2 (gloo, CPU)
_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["ARX_ROOT"])
import torch, torch.distributed as dist
from arxiv_rag_amd import config as C, generate_embeddings_parallel as GEN
from arxiv_rag_amd.index import gather_partials, shard_bounds
from arxiv_rag_amd.tokenizer import WordPieceTokenizer
from arxiv_rag_amd.weights import seeded_state_dict
from oracle import search_oracle as SO
from tests.helpers import OracleSentenceModel, synthetic_vocab
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = C.TINY_BERT
model = OracleSentenceModel(cfg, seeded_state_dict(cfg, seed=4, std=0.05), WordPieceTokenizer.from_vocab(synthetic_vocab(cfg), cfg))
os.chdir(os.environ["ARX_OUT"])
rc = GEN.main([os.environ["ARX_IN"], "--model", "tiny-bert", "--batch-size", "5", "--chunks-per-worker", "7",
               "--min-quality", "0.0", "--skip-chroma"], model_factory=lambda n: model)
assert rc == 0
# the search exchange step: local partial top-k (oracle as the stand-in for the HIP kernel) -> all_gather -> merge
Cm = SO.unit_rows_f16(1001, 64, 1); Q = SO.unit_rows_f16(9, 64, 2)
lo, hi = shard_bounds(1001, world, rank)
s, i = SO.topk_search(Cm[lo:hi], Q, 10, idx_base=lo)
all_s, all_i = gather_partials(torch.from_numpy(s), torch.from_numpy(i))
ms, mi = SO.merge_partials(all_s.numpy(), all_i.numpy(), 10)
gs, gi = SO.topk_search(Cm, Q, 10)
assert np.array_equal(mi, gi) and np.array_equal(ms, gs)
dist.barrier()
if rank == 0:
    print("WORKER_OK", sum(model.calls))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_world_size_2_gloo_matches_single_process(tmp_path, tiny_model, world):
    """One process per rank (torchrun, gloo on CPU): every rank encodes, keeps and writes its own contiguous row range; the
    three files must be the single-process run's, byte for byte; the search exchange (all_gather of partials + merge) equals the
    global answer.  world = 3 splits the 5 quanta 2 / 2 / 1 (uneven shards)."""
    cfg, sd, tok = tiny_model
    make_chunk_tree(tmp_path / "in", n_files=6, chunks_per_file=5)
    (tmp_path / "w.py").write_text(_WORKER)
    (tmp_path / "out2").mkdir(); (tmp_path / "out1").mkdir()
    env = dict(os.environ, ARX_ROOT=str(ROOT), ARX_IN=str(tmp_path / "in"), ARX_OUT=str(tmp_path / "out2"),
               OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(29615 + world), str(tmp_path / "w.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    # per-rank loading: every rank parsed its own slice of the 6 files (+ the boundary files its quanta reach into), never the tree
    parsed = {int(m.group(1)): (int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)))
              for m in re.finditer(r"\[rank (\d+)\] parsed (\d+) of 6 files \(own slice \[(\d+), (\d+)\) \+ (\d+) boundary files\)", r.stdout)}
    assert sorted(parsed) == list(range(world)), r.stdout[-3000:]
    for rk, (n, a, b, extra) in parsed.items():
        assert n == (b - a) + extra and (b - a) == len(range(*GEN.shard_range(6, world, rk)))
    assert sum(v[0] for v in parsed.values()) < world * 6
    assert not list((tmp_path / "out2" / "embeddings_saved").glob("metadata.json.part*"))      # fragments joined and removed
    cwd = os.getcwd()
    os.chdir(tmp_path / "out1")
    try:
        model = OracleSentenceModel(cfg, sd, tok)
        assert GEN.main([str(tmp_path / "in"), "--model", "tiny-bert", "--batch-size", "5", "--chunks-per-worker", "7",
                         "--min-quality", "0.0", "--skip-chroma"], model_factory=lambda n: model) == 0
    finally:
        os.chdir(cwd)
    for n in ("embeddings.npy", "metadata.json", "index.json"):
        assert (tmp_path / "out1" / "embeddings_saved" / n).read_bytes() == (tmp_path / "out2" / "embeddings_saved" / n).read_bytes(), n


_WORKER_MAIN_INIT = r"""
import os, sys
sys.path.insert(0, os.environ["ARX_ROOT"])
import torch.distributed as dist
from arxiv_rag_amd import config as C, generate_embeddings_parallel as GEN
from arxiv_rag_amd.tokenizer import WordPieceTokenizer
from arxiv_rag_amd.weights import seeded_state_dict
from tests.helpers import OracleSentenceModel, synthetic_vocab
assert not dist.is_initialized()                       # main() itself must bring the group up (its WORLD_SIZE > 1 branch)
cfg = C.TINY_BERT
model = OracleSentenceModel(cfg, seeded_state_dict(cfg, seed=4, std=0.05), WordPieceTokenizer.from_vocab(synthetic_vocab(cfg), cfg))
os.chdir(os.environ["ARX_OUT"])
rc = GEN.main([os.environ["ARX_IN"], "--model", "tiny-bert", "--batch-size", "5", "--chunks-per-worker", "7",
               "--min-quality", "0.9", "--skip-chroma"], model_factory=lambda n: model)
assert rc == 0 and dist.is_initialized() and dist.get_world_size() == int(os.environ["WORLD_SIZE"])
assert GEN.host_group() is None                        # CPU run: the default group already is gloo
dist.barrier()
if dist.get_rank() == 0:
    print("MAIN_INIT_OK")
dist.destroy_process_group()
"""


def test_main_brings_up_its_own_process_group_world_2(tmp_path, tiny_model):
    """the script's WORLD_SIZE > 1 branch end to end with NOTHING initialised by the caller — `init_distributed`
    (device bound before the group exists on a GPU box; gloo here), per-rank loading over the host group, fragment join — and the
    three files byte-equal to the one-process run's, with the quality filter dropping chunks (uneven per-file counts)."""
    cfg, sd, tok = tiny_model
    make_chunk_tree(tmp_path / "in", n_files=7, chunks_per_file=5, seed=5)
    (tmp_path / "w.py").write_text(_WORKER_MAIN_INIT)
    (tmp_path / "out2").mkdir(); (tmp_path / "out1").mkdir()
    env = dict(os.environ, ARX_ROOT=str(ROOT), ARX_IN=str(tmp_path / "in"), ARX_OUT=str(tmp_path / "out2"), OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29633", str(tmp_path / "w.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MAIN_INIT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    cwd = os.getcwd()
    os.chdir(tmp_path / "out1")
    try:
        model = OracleSentenceModel(cfg, sd, tok)
        assert GEN.main([str(tmp_path / "in"), "--model", "tiny-bert", "--batch-size", "5", "--chunks-per-worker", "7",
                         "--min-quality", "0.9", "--skip-chroma"], model_factory=lambda n: model) == 0
    finally:
        os.chdir(cwd)
    for n in ("embeddings.npy", "metadata.json", "index.json"):
        assert (tmp_path / "out1" / "embeddings_saved" / n).read_bytes() == (tmp_path / "out2" / "embeddings_saved" / n).read_bytes(), n
    src = (ROOT / "arxiv_rag_amd" / "generate_embeddings_parallel.py").read_text()
    body = src[src.index("def init_distributed"):src.index("def host_group")]
    assert body.index("torch.cuda.set_device(local_rank)") < body.index('dist.init_process_group("nccl"') and "device_id=" in body


def test_per_rank_loading_opens_only_the_files_it_needs(tmp_path, monkeypatch):
    """VERDICT r2 item 6 (GEN:94-129 loads the tree ONCE): with N ranks, rank r opens its contiguous slice of the sorted file list
    plus the files of a neighbouring slice that hold rows of its quanta — nothing else; the ranks' chunks, concatenated in rank
    order, are exactly the one-process list; the row ranges are the sharded dispatcher's; pools are sized from the cgroup quota."""
    make_chunk_tree(tmp_path / "in", n_files=23, chunks_per_file=6, seed=3)
    whole = GEN.load_chunks_parallel(tmp_path / "in", 0.9, 2)
    files = GEN.list_chunk_files(tmp_path / "in")
    per_file = [len(c) for c in GEN.load_chunk_files(files, 0.9, 2)]
    offs = np.concatenate([[0], np.cumsum(per_file)])
    assert sum(per_file) == len(whole) and 0 < len(whole) < 23 * 6              # the quality filter dropped some chunks
    real = GEN.load_chunks_from_file
    for world, cpw in ((3, 7), (8, 5), (2, 500)):
        got, n_quanta = [], (len(whole) + cpw - 1) // cpw
        for rank in range(world):
            opened = []
            monkeypatch.setattr(GEN, "load_chunks_from_file", lambda f, q, _o=opened: (_o.append(Path(f)), real(f, q))[1])
            f_lo, f_hi = GEN.shard_range(len(files), world, rank)
            counts = [[per_file[i] for i in range(*GEN.shard_range(len(files), world, r))] for r in range(world)]
            rc = GEN.load_chunks_for_rank(tmp_path / "in", 0.9, 2, cpw, world, rank, gather_counts=lambda mine, _c=counts, _r=rank: (
                _c.__setitem__(_r, mine), _c)[1])
            monkeypatch.setattr(GEN, "load_chunks_from_file", real)
            q_lo, q_hi = GEN.shard_range(n_quanta, world, rank)
            assert (rc.lo, rc.hi, rc.total) == (min(len(whole), q_lo * cpw), min(len(whole), q_hi * cpw), len(whole))
            want = set(files[f_lo:f_hi]) | {files[i] for i in range(len(files)) if per_file[i] and offs[i] < rc.hi and offs[i + 1] > rc.lo}
            assert set(opened) == want and len(opened) == len(want) == rc.files_parsed
            got.extend(rc.chunks)
        assert got == whole
    assert 1 <= GEN.default_load_workers() <= GEN.effective_cpus() <= (os.cpu_count() or 1)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert GEN.default_load_workers() == max(1, int(GEN.effective_cpus() * 0.8) // 8)


def test_batched_layout_writer_bytes_match_reference_output(tmp_path):
    """VERDICT r3 'missing' #4: the reference's ALTERNATE layout (4-embed/utils/save_embeddings_to_disk.py:15-80).  The fixture is what
    the reference's own `save_embeddings_disk` wrote for the harness chunks with batch_size = 2 (tools/make_golden.py batched); our
    writer must produce the same bytes, and our reader must read them back."""
    from arxiv_rag_amd.store import load_embeddings_from_disk, save_embeddings_disk
    fx = HARN
    chunks = []
    for f in GEN.list_chunk_files(fx / "input"):
        chunks.extend(GEN.load_chunks_from_file(f, 0.9))
    embs = np.load(fx / "input_embeddings_f32.npy")
    save_embeddings_disk(chunks, list(embs), output_dir=tmp_path / "o", batch_size=2)
    want = sorted((fx / "expected_batched").iterdir())
    assert [f.name for f in want] == sorted(f.name for f in (tmp_path / "o").iterdir())
    for f in want:
        assert (tmp_path / "o" / f.name).read_bytes() == f.read_bytes(), f.name
    e, m = load_embeddings_from_disk(tmp_path / "o")
    assert e.dtype == np.float64 and np.array_equal(e, embs.astype(np.float64)) and [x["batch_position"] for x in m] == [0, 1, 0]
    save_embeddings_disk(chunks, np.asarray(embs), output_dir=tmp_path / "o2", batch_size=10000)      # an array, one batch
    assert (tmp_path / "o2" / "index.json").read_text().count('"num_batches": 1') == 1


# ------------------------------------------------------------------ on-disk layouts -> loader
def test_load_embeddings_from_disk_both_layouts(tmp_path):
    """GEN layout (written by our writer, byte-checked above) and the batched layout of
    4-embed/utils/save_embeddings_to_disk.py:15-80 (files produced here by hand in that format)."""
    from arxiv_rag_amd.store import load_embeddings_from_disk
    rs = np.random.RandomState(0)
    chunks = [{"chunk_id": f"c{i}", "text": f"t{i}", "metadata": {"paper_id": "p", "section": "s", "quality_score": 0.95}} for i in range(25)]
    embs = rs.standard_normal((25, 8)).astype(np.float32)
    GEN.save_embeddings_to_disk_fallback(chunks, list(embs), output_dir=str(tmp_path / "gen"))
    e, m = load_embeddings_from_disk(tmp_path / "gen")
    assert e.shape == (25, 8) and e.dtype == np.float64 and np.allclose(e, embs) and [x["chunk_id"] for x in m] == [c["chunk_id"] for c in chunks]
    b = tmp_path / "batched"; b.mkdir()
    for i in range(3):
        lo, hi = i * 10, min(25, i * 10 + 10)
        np.save(b / f"embeddings_batch_{i:04d}.npy", embs[lo:hi].astype(np.float64))
        (b / f"metadata_batch_{i:04d}.json").write_text(json.dumps(
            [{"chunk_id": f"c{j}", "text": f"t{j}", "batch_index": i, "batch_position": j - lo} for j in range(lo, hi)]))
    (b / "index.json").write_text(json.dumps({"total_embeddings": 25, "embedding_dimension": 8, "num_batches": 3, "batch_size": 10,
                                              "chunks": [f"c{j}" for j in range(25)]}))
    e2, m2 = load_embeddings_from_disk(b)
    assert np.allclose(e2, embs) and [x["chunk_id"] for x in m2] == [f"c{j}" for j in range(25)]
    e3, m3 = load_embeddings_from_disk(b, batch_index=2)
    assert e3.shape == (5, 8) and m3[0]["chunk_id"] == "c20"


def test_semantic_grouping_matches_reference_walk():
    """group_sentences / split_sentences against chunks produced by the reference's own loop
    (text_processor.py:1275-1276, 1542-1599; fixture from tools/make_golden.py `semantic`)."""
    from arxiv_rag_amd.semantic import group_sentences, split_sentences
    fx = json.loads((Path(__file__).parent / "golden" / "semantic_chunker.json").read_text(encoding="utf-8"))
    assert split_sentences(fx["split"]["text"]) == fx["split"]["expected"]
    assert len(fx["cases"]) >= 7
    for c in fx["cases"]:
        got = group_sentences(c["sentences"], c["similarities"], c["max_chunk_size"], c["min_chunk_size"], c["metadata"])
        assert got == c["expected_chunks"]
    assert group_sentences([], []) == []
    with pytest.raises(ValueError):
        group_sentences(["a", "b"], [])


def test_bench_starts_its_own_ranks_when_called_plainly():
    """`python bench.py --gpus N` (how the driver calls it, no RANK in the environment) must launch N ranks itself; under
    torch.distributed.run (RANK set) and for N = 1 it must not."""
    import bench
    assert bench.needs_self_launch(8, {}) and bench.needs_self_launch(2, {"WORLD_SIZE": "1"})
    assert not bench.needs_self_launch(1, {})
    assert not bench.needs_self_launch(8, {"RANK": "3", "WORLD_SIZE": "8"})
    assert bench.needs_self_launch(1, {"ARX_BENCH_FORCE_LAUNCH": "1"})
    cmd = bench.relaunch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]          # the ranks get the caller's arguments unchanged


def test_committed_bench_line_honours_the_contract():
    """The bench line committed under profiles/ (what `python bench.py` printed on the GPU box) carries every key the driver
    and the judge read, with consistent arithmetic."""
    d = json.loads((ROOT / "profiles" / "r04_bench_final.json").read_text().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "chunks/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    per_step = d["config"]["global_batch"] * 1000.0 / d["ms_per_step"]
    assert abs(per_step - d["value"]) / d["value"] < 0.02                       # value = chunks per step / time per step
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    legs = {k: v for k, v in c["legs"].items() if "value" in v}                 # A = one process x all usable cores, C = GEN:190 fan-out
    assert set(legs) == {"A", "C"} and c["value"] == max(v["value"] for v in legs.values())
    assert c["cores"] == max(legs.values(), key=lambda v: v["value"])["processes"] * max(legs.values(), key=lambda v: v["value"])["threads_per_process"]
    assert legs["C"]["processes"] == int(0.75 * c["host"]["usable_cpus"]) and legs["C"]["threads_per_process"] == 1
    su = d["encode"]["sustained"]                                               # the whole configs[1] job, not a burst
    assert su["chunks"] == 1_000_000 and su["batches_per_gpu"] == 977
    assert abs(su["chunks"] / su["seconds"] - su["chunks_per_s"]) / su["chunks_per_s"] < 0.01
    assert abs(su["ms_per_step_last_100"] / su["ms_per_step_first_100"] - 1) < 0.05          # no clock droop in the record
    s = d["search"]["roofline"]
    assert s["bound"] == "hbm" and abs(s["achieved"] / s["peak"] - s["frac"]) < 1e-3
    # round 4: the clock the cap allowed is in the line, with the peak and the fraction it implies; the new legs are there
    assert 1500 < r["clock_mhz_under_load"] <= 2400 and abs(r["peak_at_measured_clock"] - r["peak"] * r["clock_mhz_under_load"] / 2400) < 1.0
    assert abs(r["achieved"] / r["peak_at_measured_clock"] - r["frac_of_peak_at_measured_clock"]) < 2e-3
    assert d["launch"]["ranks_seen"] == list(range(d["n_gpus"])) and d["launch"]["gpus_visible"] >= d["launch"]["local_world_size"]
    assert d["encode"]["bge_large"]["chunks_per_s"] > 0 and d["e2e_rank_slice"]["vs_oracle"]["top10_sets_equal_oracle"] is True
    k625 = d["search"]["shard_625k"]["results"]["Qb=64"]
    assert abs(k625["passA_ms_at_hbm_roofline"] / k625["ms_per_batch_pipelined"] - k625["batch_frac_of_hbm_roofline_pipelined"]) < 2e-3
    assert d["search"]["clustered"]["top10_scores_equal_fp32_reference_on_8_queries"] is True
