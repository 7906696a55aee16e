"""GPU tests (-m gpu) of the drop-in script on the HIP backend, through the code a user of the reference hits:
`--model <name>` resolved to a LOCAL directory (hub.py; GEN:47, GEN:474), weights from `model.safetensors`, tokenizer from
`vocab.txt` / `tokenizer.json`, `--batch-size 32` (BASELINE.json configs[0] at its stated shape), the added `--queries`
step, and the multi-rank branch with a real RCCL group (1 rank: `all_gather_into_tensor` + `arx_topk_merge`)."""
import json
import os

import numpy as np
import pytest

from arxiv_rag_amd import config as C
from arxiv_rag_amd.weights import seeded_state_dict
from oracle import encoder_oracle as EO
from oracle import search_oracle as SO

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1) + 1e-30)


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from arxiv_rag_amd import _lib
    _lib.load()
    return _lib


def _minilm_model_dir(root, tokenizer_json=False):
    """all-MiniLM-L6-v2 SHAPE (6L/384, 30522-entry vocabulary) as a local HF-layout directory: seeded weights in
    model.safetensors, a synthetic WordPiece vocabulary padded with [unusedN] to the checkpoint's vocab size."""
    from arxiv_rag_amd.tokenizer import WordPieceTokenizer
    from arxiv_rag_amd.weights import save_hf_dir
    from tests.helpers import synthetic_vocab
    cfg = C.MINILM_L6
    sd = seeded_state_dict(cfg, seed=3, std=0.04, bias_std=0.02, ln_jitter=0.05)
    mdir = root / "models" / "all-MiniLM-L6-v2"
    save_hf_dir(mdir, cfg, sd)
    vocab = synthetic_vocab(C.EncoderConfig(**{**cfg.__dict__, "vocab_size": 2000}))
    toks = sorted(vocab, key=vocab.get)
    toks += [f"[unused{i}]" for i in range(cfg.vocab_size - len(toks))]
    if tokenizer_json:
        full = {t: i for i, t in enumerate(toks)}
        WordPieceTokenizer.from_vocab(full, cfg)._tok.save(str(mdir / "tokenizer.json"))
    else:
        (mdir / "vocab.txt").write_text("\n".join(toks) + "\n", encoding="utf-8")
    words = [w for w in vocab if w.isalpha() and len(w) > 1][:300]
    return cfg, sd, mdir, words


@pytest.mark.parametrize("tokenizer_json", [False, True])
def test_cli_drop_in_end_to_end_on_gpu(hip, tmp_path, monkeypatch, tokenizer_json):
    """BASELINE configs[0] as a parity case: 1 000 pre-chunked JSON docs, all-MiniLM-L6-v2 SHAPE (384-d, seeded
    weights saved as a local HF-layout directory with a synthetic WordPiece vocab), --batch-size 32, through the
    drop-in script on the HIP backend with NO injected model factory; rows must match the CPU oracle on the same token
    ids (cosine >= 1-1e-3), layout/dtype/order as the reference writes them, and the added --queries step (over the fp16
    shard the encode step left in HBM) must agree with the search oracle."""
    from arxiv_rag_amd import generate_embeddings_parallel as GEN
    from arxiv_rag_amd.tokenizer import WordPieceTokenizer
    from tests.helpers import make_chunk_tree
    cfg, sd, mdir, words = _minilm_model_dir(tmp_path, tokenizer_json)
    make_chunk_tree(tmp_path / "in", n_files=100, chunks_per_file=10, seed=1, words=words)       # 1 000 chunks
    (tmp_path / "queries.txt").write_text("\n".join(" ".join(words[i:i + 6]) for i in range(0, 60, 6)) + "\n")
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    GEN._model, GEN._model_name = None, None
    rc = GEN.main([str(tmp_path / "in"), "--model", "all-MiniLM-L6-v2", "--model-dir", str(tmp_path / "models"),
                   "--batch-size", "32", "--min-quality", "0.9", "--skip-chroma", "--queries", str(tmp_path / "queries.txt")])
    assert rc == 0
    kept = GEN.load_chunks_parallel(tmp_path / "in", 0.9, 4)
    arr = np.load(tmp_path / "embeddings_saved" / "embeddings.npy")
    assert arr.dtype == np.float64 and arr.shape == (len(kept), 384) and 300 < len(kept) < 1000
    meta = json.loads((tmp_path / "embeddings_saved" / "metadata.json").read_text())
    assert [m["chunk_id"] for m in meta] == [c["chunk_id"] for c in kept]
    idx = json.loads((tmp_path / "embeddings_saved" / "index.json").read_text())
    assert idx["total_embeddings"] == len(kept) and idx["embedding_dimension"] == 384
    tok = WordPieceTokenizer.from_dir(mdir, cfg)
    seqs = tok.encode_batch([c["text"] for c in kept], cfg.max_seq_length)
    ref = EO.encode_ragged(sd, cfg, seqs, batch_size=64)
    assert _cos(arr.astype(np.float32), ref).min() > 1 - 1e-3
    assert np.abs(np.linalg.norm(arr, axis=1) - 1).max() < 1e-4
    # the search step over the fp16 rows the script kept in HBM
    res = json.loads((tmp_path / "embeddings_saved" / "search_results.json").read_text())
    qs = (tmp_path / "queries.txt").read_text().split("\n")[:-1]
    assert [r["query"] for r in res] == qs
    qref = EO.encode_ragged(sd, cfg, tok.encode_batch(qs, cfg.max_seq_length))
    rs, ri = SO.topk_search(arr.astype(np.float16), qref.astype(np.float16), 10)
    hit = 0
    for qi, r in enumerate(res):
        got = [h["index"] for h in r["results"]]
        assert [h["chunk_id"] for h in r["results"]] == [kept[j]["chunk_id"] for j in got]
        hit += len(set(got) & set(ri[qi].tolist()))
    assert hit >= 0.95 * 10 * len(qs)          # bf16 query/corpus rows vs fp32-oracle rows: near-ties may swap
    # exact statement of the same step: top-10 over the fp16 rows the kernels wrote, queries as the kernels encoded them
    model = GEN._model
    qd = torch.empty((len(qs), 384), dtype=torch.float16, device="cuda")
    model.encode(qs, normalize_embeddings=True, device_f16_out=qd, low_latency=True)   # as search_queries encodes them
    rs2, ri2 = SO.topk_search(arr.astype(np.float16), qd.cpu().numpy(), 11)
    for qi, r in enumerate(res):
        got = [h["index"] for h in r["results"]]
        if set(got) != set(ri2[qi, :10].tolist()):
            assert rs2[qi, 9] - rs2[qi, 10] < 1e-6
    GEN._model, GEN._model_name = None, None


def test_hub_resolution_and_shape_check(hip, tmp_path, monkeypatch):
    """hub.resolve_model_dir: directory itself, $ARX_MODEL_DIR/<name>, org-prefixed name; a wrong-shape checkpoint under a
    preset's name is refused; a missing model is FileNotFoundError (the CLI's exit code 1), never a download."""
    from arxiv_rag_amd import generate_embeddings_parallel as GEN
    from arxiv_rag_amd.hub import load_sentence_encoder, resolve_model_dir
    from arxiv_rag_amd.weights import save_hf_dir
    from tests.helpers import synthetic_vocab
    cfg = C.TINY_BERT
    sd = seeded_state_dict(cfg, seed=4, std=0.05)
    d = tmp_path / "zoo" / "tiny-bert"
    save_hf_dir(d, cfg, sd)
    vocab = synthetic_vocab(cfg)
    (d / "vocab.txt").write_text("\n".join(sorted(vocab, key=vocab.get)) + "\n", encoding="utf-8")
    assert resolve_model_dir(str(d)) == d
    monkeypatch.setenv("ARX_MODEL_DIR", str(tmp_path / "zoo"))
    assert resolve_model_dir("tiny-bert") == d and resolve_model_dir("some-org/tiny-bert") == d
    m = load_sentence_encoder("tiny-bert")
    texts = ["ab cd ef", "a", "zz yy xx ww"]
    rows = m.encode(texts, normalize_embeddings=True)
    ref = EO.encode_ragged(sd, cfg, m.tokenize(texts))
    assert _cos(rows, ref).min() > 1 - 1e-3
    m.encoder.close()
    # a tiny checkpoint filed under a preset's name: shape mismatch is an error, not a silent mis-load
    bad = tmp_path / "zoo" / "all-mpnet-base-v2"
    save_hf_dir(bad, cfg, sd)
    (bad / "vocab.txt").write_text((d / "vocab.txt").read_text())
    with pytest.raises(ValueError):
        load_sentence_encoder("all-mpnet-base-v2")
    with pytest.raises(FileNotFoundError):
        resolve_model_dir("all-MiniLM-L6-v2")
    (tmp_path / "in").mkdir()
    (tmp_path / "in" / "x.json").write_text(json.dumps({"chunks": [{"chunk_id": "c", "text": "ab", "metadata": {"quality_score": 1.0}}]}))
    monkeypatch.chdir(tmp_path)
    GEN._model, GEN._model_name = None, None
    assert GEN.main([str(tmp_path / "in"), "--model", "all-MiniLM-L6-v2", "--skip-chroma"]) == 1
    GEN._model, GEN._model_name = None, None


# ---------------------------------------------------------------------------------------------- RCCL, one rank
@pytest.fixture(scope="module")
def nccl_world1(hip):
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                device_id=torch.device("cuda:0"))
        created = True
    yield dist
    if created:
        dist.destroy_process_group()


def test_search_distributed_single_rank_nccl(nccl_world1):
    """The RCCL exchange step with a 1-rank group on the GPU: all_gather_into_tensor + merge kernel == local search."""
    from arxiv_rag_amd.index import ShardIndex
    Cm = SO.unit_rows_f16(5000, 128, 1); Q = SO.unit_rows_f16(33, 128, 2)
    idx = ShardIndex(torch.from_numpy(Cm).cuda(), idx_base=7)
    s0, i0 = idx.search(torch.from_numpy(Q).cuda(), 10)
    s1, i1 = idx.search_distributed(torch.from_numpy(Q).cuda(), 10)
    assert torch.equal(i0, i1) and torch.equal(s0, s1)
    rs, ri = SO.topk_search(Cm, Q, 10)
    assert np.array_equal(i1.cpu().numpy() - 7, ri)
    # the pipelined form (two batches in flight, all-gather + merge on the side streams)
    qd = torch.from_numpy(Q).cuda()
    got = idx.search_many([qd[:16], qd[16:32], qd[32:]], 10, distributed=True)
    assert torch.equal(torch.cat([i for _, i in got]), i1) and torch.equal(torch.cat([s for s, _ in got]), s1)


def test_sharded_branch_single_rank_nccl(nccl_world1, tmp_path, monkeypatch):
    """The multi-rank branch of the script (`generate_embeddings_sharded` + `save_embeddings_sharded` +
    `search_queries` over the rank's HBM shard) under a real NCCL(=RCCL) process group of one rank: same files and the same
    hits as the single-process branch, and the shard the search runs on is the one the encoder kernels wrote."""
    from arxiv_rag_amd import generate_embeddings_parallel as GEN
    from arxiv_rag_amd.hub import load_sentence_encoder
    from tests.helpers import make_chunk_tree
    cfg, sd, mdir, words = _minilm_model_dir(tmp_path)
    make_chunk_tree(tmp_path / "in", n_files=30, chunks_per_file=10, seed=2, words=words)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("ARX_MODEL_DIR", str(tmp_path / "models"))
    chunks = GEN.load_chunks_parallel(tmp_path / "in", 0.85, 2)
    qs = [" ".join(words[i:i + 5]) for i in range(0, 40, 5)]
    GEN._model, GEN._model_name = None, None
    model = GEN.init_worker_model("all-MiniLM-L6-v2")
    assert GEN._dist() is not None and GEN._dist().get_world_size() == 1
    # single-process dispatcher (reference order contract) as the expectation
    base = np.asarray(GEN.generate_embeddings_parallel(chunks, "all-MiniLM-L6-v2", batch_size=32, chunks_per_worker=50), np.float32)
    sink = GEN.make_shard_sink(model, len(chunks), 50, 1, 0)
    rows, lo, hi = GEN.generate_embeddings_sharded(chunks, "all-MiniLM-L6-v2", batch_size=32, chunks_per_worker=50, sink=sink)
    assert (lo, hi) == (0, len(chunks)) and np.array_equal(rows, base)
    assert torch.equal(sink.rows.cpu(), torch.from_numpy(rows.astype(np.float16)))          # the kernels' own fp16 rows
    GEN.save_embeddings_sharded(chunks, rows, lo, hi, output_dir=str(tmp_path / "embeddings_saved"))
    arr = np.load(tmp_path / "embeddings_saved" / "embeddings.npy")
    assert arr.dtype == np.float64 and np.array_equal(arr, rows.astype(np.float64))
    meta = json.loads((tmp_path / "embeddings_saved" / "metadata.json").read_text())
    assert [m["chunk_id"] for m in meta] == [c["chunk_id"] for c in chunks]
    r_dev = GEN.search_queries(model, chunks, sink, qs, top_k=10, output_dir=str(tmp_path / "embeddings_saved"))
    qd = torch.empty((len(qs), 384), dtype=torch.float16, device="cuda")
    model.encode(qs, normalize_embeddings=True, device_f16_out=qd, low_latency=True)   # as search_queries encodes them
    # results unchanged by where the shard came from: the same search over the rows of the .npy file, uploaded here by the test
    from arxiv_rag_amd.index import ShardIndex
    s_up, i_up = ShardIndex(torch.from_numpy(arr.astype(np.float16)).cuda(), idx_base=lo).search_distributed(qd, 10)
    assert [[h["index"] for h in r["results"]] for r in r_dev] == i_up.cpu().tolist()
    assert np.array_equal(np.array([[h["score"] for h in r["results"]] for r in r_dev], np.float32), s_up.cpu().numpy())
    rs, ri = SO.topk_search(rows.astype(np.float16), qd.cpu().numpy(), 11)
    for qi, r in enumerate(r_dev):
        got = [h["index"] for h in r["results"]]
        if set(got) != set(ri[qi, :10].tolist()):
            assert rs[qi, 9] - rs[qi, 10] < 1e-6
    # a failed quantum (reference policy: zero rows) must also reach the HBM shard
    real = model.encode

    def flaky(sentences, **kw):
        if any("FAILME" in t for t in sentences):
            raise RuntimeError("injected")
        return real(sentences, **kw)
    bad = [dict(c) for c in chunks]
    bad[60] = {**bad[60], "text": "FAILME " + bad[60]["text"]}
    model.encode = flaky
    try:
        sink2 = GEN.make_shard_sink(model, len(bad), 50, 1, 0)
        rows2, _, _ = GEN.generate_embeddings_sharded(bad, "all-MiniLM-L6-v2", batch_size=32, chunks_per_worker=50, sink=sink2)
    finally:
        model.encode = real
    assert not rows2[60].any() and np.array_equal(np.delete(rows2, 60, 0), np.delete(rows, 60, 0))
    assert torch.equal(sink2.rows.cpu(), torch.from_numpy(rows2.astype(np.float16)))
    model.encoder.close()
    GEN._model, GEN._model_name = None, None


def test_bench_starts_its_own_rank_and_prints_one_json_line(hip):
    """`python bench.py --gpus N` as the driver calls it (no RANK in the environment) must start its ranks itself as child processes
    and still print exactly ONE JSON line.  Rehearsed with the one GPU there is: ARX_BENCH_FORCE_LAUNCH=1 sends N = 1 down the same
    child-launch path (torch.distributed.run, RCCL group of one rank, all-gather + merge of the partial top-k); small shapes."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ARX_BENCH_FORCE_LAUNCH"] = "1"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--sustained-chunks", "4096", "--e2e-rows", "3000", "--clustered-rows", "0", "--bge-steps", "0",
                        "--no-cpu-baseline", "--no-query-leg", "--search-rows", "200000",
                        "--search-total-rows", "100000", "--d1024-rows", "0", "--search-queries", "300"],
                       env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
    st = d["search"]["strong_scaling"]
    assert st["merged_vs_single_index"] == {"queries": 64, "ids_equal": True, "scores_equal": True}
    assert "allgather_plus_merge_ms" in st["results"]["Qb=64"] and st["results"]["Qb=64"]["qps_pipelined"] > 0
    # round 4: the launch self-checks ran under the real process group, and the rows the encoder wrote were searched through it
    assert d["launch"]["backend"] == "nccl" and d["launch"]["ranks_seen"] == [0] and d["launch"]["row_spans"] == [[0, 100000]]
    e2e = d["e2e_rank_slice"]
    assert "error" not in e2e and e2e["vs_oracle"]["top10_sets_equal_oracle"] is True and e2e["results"]["Qb=64"]["qps_pipelined"] > 0


def test_cli_init_distributed_nccl_with_gloo_host_group(hip, tmp_path):
    """ADVICE r3 (high), the part a CPU test cannot reach: `init_distributed()` on a GPU box — device bound BEFORE the RCCL group exists,
    `device_id` passed — and `host_group()`: a gloo group beside the nccl one, carrying the loader's `all_gather_object` without touching
    device memory.  One rank (the only GPU here), in a child process with a clean environment; then the script's whole multi-rank branch
    runs under it (per-rank loading -> encode -> fragment join)."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    from tests.helpers import make_chunk_tree
    make_chunk_tree(tmp_path / "in", n_files=5, chunks_per_file=4, seed=2)
    code = r"""
import os, sys
sys.path.insert(0, os.environ["ARX_ROOT"])
import torch, torch.distributed as dist
from arxiv_rag_amd import generate_embeddings_parallel as GEN
assert not dist.is_initialized()
GEN.init_distributed()
assert dist.get_backend() == "nccl" and torch.cuda.current_device() == 0
hg = GEN.host_group()
assert hg is not None and dist.get_backend(hg) == "gloo"
out = [None]
dist.all_gather_object(out, {"files": [3, 1, 4]}, group=hg)
assert out == [{"files": [3, 1, 4]}]
rc = GEN.load_chunks_for_rank(os.environ["ARX_IN"], 0.0, 2, 7, 1, 0)          # the count exchange over the host group
assert (rc.lo, rc.hi, rc.total) == (0, 20, 20)
t = torch.ones(4, device="cuda"); dist.all_reduce(t); assert t.sum().item() == 4   # and the nccl group works
dist.barrier(); dist.destroy_process_group()
print("INIT_OK")
"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ARX_ROOT=str(root), ARX_IN=str(tmp_path / "in"), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "INIT_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
