#!/usr/bin/env python3
"""Generate tests/golden/* — run in the BUILD container only (needs `transformers`, and
/root/reference for the writer/loader fixtures).  Nothing here runs on the GPU box.

Sources of truth:
  * encoder vectors  <- transformers.MPNetModel / BertModel (eager attention, fp32 CPU), built
    from explicit local configs with weights from arxiv_rag_amd.weights.seeded_state_dict, then
    the two sentence-transformers tail modules (mean/CLS pool, L2 normalise) in torch.
  * relative-position buckets / position ids <- the transformers functions themselves.
  * loader / writer bytes <- the reference's own functions, extracted BY LINE RANGE from
    /root/reference/4-embed/generation/generate_embeddings_parallel.py (the file as a whole has a
    SyntaxError at line 239 and cannot be imported; the two functions used parse on their own)
    and executed on the fixture inputs.  Only their OUTPUT is written here.
  * search vectors <- oracle/search_oracle.py cross-checked against torch.topk on fp32 CPU matmul
    (no reference search code exists; "parity unpinned").
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from arxiv_rag_amd import config as C                      # noqa: E402
from arxiv_rag_amd.weights import seeded_state_dict        # noqa: E402

GOLD = ROOT / "tests" / "golden"
GEN = Path("/root/reference/4-embed/generation/generate_embeddings_parallel.py")


def build_tf_model(cfg: C.EncoderConfig, sd):
    from oracle import tf_reference as TF
    return TF.build_model(cfg, sd)


def tf_encode(m, cfg, ids, lens, all_hidden=False):
    ids_t = torch.from_numpy(ids)
    S = ids.shape[1]
    mask = (torch.arange(S)[None, :] < torch.from_numpy(lens)[:, None]).long()
    with torch.no_grad():
        out = m(input_ids=ids_t, attention_mask=mask, output_hidden_states=all_hidden, return_dict=True)
    h = out.last_hidden_state
    if cfg.pool == C.POOL_CLS:
        pooled = h[:, 0]
    else:   # sentence_transformers.models.Pooling (mean)
        mf = mask.unsqueeze(-1).float()
        pooled = (h * mf).sum(1) / torch.clamp(mf.sum(1), min=1e-9)
    emb = torch.nn.functional.normalize(pooled, p=2, dim=1)
    hs = [x.numpy() for x in out.hidden_states] if all_hidden else None
    return h.numpy(), pooled.numpy(), emb.numpy(), hs


def ragged_ids(cfg, lens, seed):
    """Right-padded token ids: MPNet <s>=0 ... </s>=2 pad=1; BERT [CLS]=101%V ... [SEP]=102%V pad=0."""
    rs = np.random.RandomState(seed)
    S = int(max(lens))
    ids = np.full((len(lens), S), cfg.pad_id, np.int64)
    lo = 4
    for r, n in enumerate(lens):
        body = rs.randint(lo, cfg.vocab_size - 1, size=n)
        if cfg.arch == C.ARCH_MPNET:
            body[0] = 0
            if n > 1:
                body[-1] = 2
        else:
            body[0] = 101 % cfg.vocab_size
            if n > 1:
                body[-1] = 102 % cfg.vocab_size
        ids[r, :n] = body
    return ids


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode()); h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()


WSPEC = dict(std=0.05, bias_std=0.05, ln_jitter=0.1)     # peaky attention, non-trivial LN affine


def gen_tiny():
    for name, lens in (("tiny-mpnet", [64, 33, 17, 8, 3, 1, 40, 64]),
                       ("tiny-bert", [64, 31, 16, 9, 2, 1, 50, 7]),
                       ("tiny-bert-cls", [48, 20, 5, 1])):
        cfg = C.PRESETS[name]
        sd = seeded_state_dict(cfg, seed=11, **WSPEC)
        m = build_tf_model(cfg, sd)
        lens = np.array(lens, np.int64)
        ids = ragged_ids(cfg, lens, seed=5)
        h, pooled, emb, hs = tf_encode(m, cfg, ids, lens, all_hidden=True)
        arrs = {"ids": ids, "lens": lens, "pooled": pooled, "emb": emb}
        for i, x in enumerate(hs):
            arrs[f"hidden_{i}"] = x
        for k, v in sd.items():
            arrs["w:" + k] = v
        np.savez_compressed(GOLD / f"{name}.npz", **arrs)
        print(name, "emb", emb.shape, "weights digest", sd_digest(sd)[:12])


def gen_full():
    specs = (
        ("all-mpnet-base-v2", [256, 200, 128, 64, 32, 16, 8, 3, 255, 129, 97, 77, 65, 33, 1, 256]),
        ("all-MiniLM-L6-v2", [256, 200, 128, 64, 32, 16, 8, 3, 255, 129, 97, 77, 65, 33, 1, 256]),
        ("BAAI/bge-large-en-v1.5", [256, 100, 37, 5]),
    )
    out = {}
    for name, lens in specs:
        cfg = C.PRESETS[name]
        for tag, wspec, seed in (("w05", WSPEC, 0), ("hf02", dict(std=0.02, bias_std=0.02, ln_jitter=0.0), 0)):
            if tag == "hf02" and cfg.layers > 12:
                continue
            sd = seeded_state_dict(cfg, seed=seed, **wspec)
            m = build_tf_model(cfg, sd)
            lens_a = np.array(lens, np.int64)
            ids = ragged_ids(cfg, lens_a, seed=1234)
            _, _, emb, _ = tf_encode(m, cfg, ids, lens_a)
            key = name.replace("/", "_") + ":" + tag
            out[key + ":emb"] = emb.astype(np.float32)
            out[key + ":ids"] = ids
            out[key + ":lens"] = lens_a
            out[key + ":wspec"] = np.array([seed, wspec["std"], wspec["bias_std"], wspec["ln_jitter"]], np.float64)
            out[key + ":wdigest"] = np.frombuffer(bytes.fromhex(sd_digest(sd)), np.uint8)
            print(key, emb.shape, float(np.abs(emb).max()))
            del m, sd
    np.savez_compressed(GOLD / "full_shapes.npz", **out)


def gen_adversarial():
    """full_shapes_adv.npz: the full shapes again with `weights.adversarial_state_dict` (wide LayerNorm affine with outlier
    channels, rows whose mean dwarfs their spread, heavy-tailed matrices) — the LN-fold path's worst case."""
    from arxiv_rag_amd.weights import adversarial_state_dict
    specs = (("all-mpnet-base-v2", [256, 200, 128, 64, 32, 16, 8, 3, 255, 129, 97, 77, 65, 33, 1, 256], 4.0),
             ("all-MiniLM-L6-v2", [256, 200, 128, 64, 32, 16, 8, 3, 255, 129, 97, 77, 65, 33, 1, 256], 4.0),
             ("BAAI/bge-large-en-v1.5", [256, 100, 37, 5], 4.0))
    out = {}
    for name, lens, off in specs:
        cfg = C.PRESETS[name]
        sd = adversarial_state_dict(cfg, seed=3, row_offset=off)
        m = build_tf_model(cfg, sd)
        lens_a = np.array(lens, np.int64)
        ids = ragged_ids(cfg, lens_a, seed=4321)
        h, _, emb, hs = tf_encode(m, cfg, ids, lens_a, all_hidden=True)
        key = name.replace("/", "_") + ":adv"
        out[key + ":emb"] = emb.astype(np.float32)
        out[key + ":ids"] = ids
        out[key + ":lens"] = lens_a
        out[key + ":wspec"] = np.array([3, off], np.float64)
        out[key + ":wdigest"] = np.frombuffer(bytes.fromhex(sd_digest(sd)), np.uint8)
        valid = np.arange(ids.shape[1])[None] < lens_a[:, None]
        x = hs[-1][valid]
        print(key, emb.shape, "last hidden: mean |row mean| %.2f, mean row std %.2f, max |x| %.1f" %
              (np.abs(x.mean(1)).mean(), x.std(1).mean(), np.abs(x).max()))
        del m, sd
    np.savez_compressed(GOLD / "full_shapes_adv.npz", **out)


def gen_tables():
    from transformers.models.mpnet.modeling_mpnet import MPNetEncoder, create_position_ids_from_input_ids
    out = {}
    for S in (8, 256, 384, 512):
        ctx = torch.arange(S)[:, None]; mem = torch.arange(S)[None, :]
        out[f"bucket_S{S}_row0"] = MPNetEncoder.relative_position_bucket(mem - ctx)[0].numpy()
        out[f"bucket_S{S}_rowlast"] = MPNetEncoder.relative_position_bucket(mem - ctx)[-1].numpy()
    d = torch.arange(-600, 601)
    out["bucket_of_delta"] = MPNetEncoder.relative_position_bucket(d).numpy()
    out["delta"] = d.numpy()
    ids = torch.tensor([[0, 5, 6, 2, 1, 1], [0, 9, 2, 1, 1, 1], [0, 7, 8, 9, 10, 2], [1, 1, 1, 1, 1, 1]])
    out["posid_ids"] = ids.numpy()
    out["posid_expected"] = create_position_ids_from_input_ids(ids, 1).numpy()
    np.savez_compressed(GOLD / "mpnet_tables.npz", **out)


def gen_search():
    sys.path.insert(0, str(ROOT))
    from oracle.search_oracle import topk_search, unit_rows_f16
    Cm = unit_rows_f16(4096, 768, seed=7)
    Q = unit_rows_f16(64, 768, seed=11)
    # constructed exact ties: rows 100, 2000, 3000 are copies of row 17; query 0 == row 17
    Cm[100] = Cm[17]; Cm[2000] = Cm[17]; Cm[3000] = Cm[17]; Q[0] = Cm[17]
    s, i = topk_search(Cm, Q, 10)
    S = torch.from_numpy(Q.astype(np.float32)) @ torch.from_numpy(Cm.astype(np.float32)).T
    ts, ti = torch.topk(S, 10, dim=1)
    for q in range(64):
        assert set(ti[q].tolist()) == set(i[q].tolist()), q
    assert i[0, :4].tolist() == [17, 100, 2000, 3000]
    np.savez_compressed(GOLD / "search_4096x768.npz", scores=s, ids=i,
                        recipe=np.array([4096, 768, 7, 64, 11], np.int64))
    print("search golden ok; q0 ids", i[0].tolist())


def _extract(src_lines, first, last):
    return "".join(src_lines[first - 1:last])


def gen_harness():
    """Run the reference's own loader (GEN:76-92) and writer (GEN:271-321) on 3 fixture files."""
    if not GEN.exists():
        print("reference absent; skipping harness fixtures"); return
    lines = GEN.read_text().splitlines(keepends=True)
    ns = {"json": json, "np": np, "Path": Path, "List": list, "Dict": dict}
    exec(compile(_extract(lines, 76, 92), "GEN:76-92", "exec"), ns)
    exec(compile(_extract(lines, 271, 321), "GEN:271-321", "exec"), ns)
    load_fn, save_fn = ns["load_chunks_from_file"], ns["save_embeddings_to_disk_fallback"]

    fx = GOLD / "harness"
    (fx / "input" / "sub").mkdir(parents=True, exist_ok=True)
    files = {
        "input/paper_a.json": {"paper_id": "0704.0001", "chunks": [
            {"chunk_id": "0704.0001_chunk_0", "text": "alpha beta gamma delta",
             "metadata": {"quality_score": 0.95, "paper_id": "0704.0001", "section": "Introduction", "chunk_index": 0}},
            {"chunk_id": "0704.0001_chunk_1", "text": "low quality chunk dropped by the filter",
             "metadata": {"quality_score": 0.5, "paper_id": "0704.0001", "section": "Methods", "chunk_index": 1}},
            {"chunk_id": "0704.0001_chunk_2", "text": "épsilon ζeta — unicode survives ensure_ascii=False",
             "metadata": {"quality_score": 0.9, "paper_id": "0704.0001", "section": None, "chunk_index": 2}}]},
        "input/sub/paper_b.json": {"paper_id": "0704.0002", "chunks": [
            {"text": "chunk without an id gets chunk_<i>",
             "metadata": {"quality_score": 1.0, "paper_id": "0704.0002", "section": "Results"}},
            {"chunk_id": "0704.0002_chunk_1", "text": "no quality score means zero", "metadata": {"paper_id": "0704.0002"}}]},
        "input/._hidden.json": {"chunks": [{"chunk_id": "x", "text": "resource fork file must be skipped",
                                            "metadata": {"quality_score": 1.0}}]},
        "input/broken.json": None,
    }
    for rel, obj in files.items():
        p = fx / rel
        p.write_text("{ not json" if obj is None else json.dumps(obj, ensure_ascii=False, indent=1), encoding="utf-8")

    # reference loader, per file, deterministic (sorted) order — imap_unordered order is not a contract
    all_files = sorted(f for f in (fx / "input").rglob("*.json") if not f.name.startswith("._"))
    chunks = []
    for f in all_files:
        chunks.extend(load_fn(f, min_quality=0.9))
    (fx / "expected_loaded_ids.json").write_text(json.dumps(
        [c.get("chunk_id") for c in chunks]))
    rs = np.random.RandomState(3)
    embs = [rs.standard_normal(8).astype(np.float32) for _ in chunks]
    embs[1] = np.zeros(8)                        # the zero-vector fallback row is float64 (GEN:169)
    np.save(fx / "input_embeddings_f32.npy", np.stack([e.astype(np.float32) for e in embs]))
    with tempfile.TemporaryDirectory() as td:
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            save_fn(chunks, embs, output_dir=td)
        for n in ("metadata.json", "index.json"):
            (fx / ("expected_" + n)).write_bytes((Path(td) / n).read_bytes())
        arr = np.load(Path(td) / "embeddings.npy")
        (fx / "expected_embeddings_meta.json").write_text(json.dumps(
            {"dtype": str(arr.dtype), "shape": list(arr.shape), "fortran_order": bool(np.isfortran(arr)),
             "sha256": hashlib.sha256(arr.tobytes()).hexdigest()}))
        (fx / "expected_embeddings.npy").write_bytes((Path(td) / "embeddings.npy").read_bytes())
    print("harness fixtures:", len(chunks), "chunks kept")


def gen_harness_batched():
    """The reference's ALTERNATE on-disk layout: its own `save_embeddings_disk` (4-embed/utils/save_embeddings_to_disk.py:15-80, extracted
    by line range and run HERE) on the harness chunks with batch_size = 2 -> tests/golden/harness/expected_batched/* (two batches)."""
    src = GEN.parents[1] / "utils" / "save_embeddings_to_disk.py"
    if not src.exists() or not (GOLD / "harness" / "input_embeddings_f32.npy").exists():
        print("reference or harness fixtures absent; skipping the batched-layout fixture"); return
    from tqdm import tqdm
    lines = src.read_text().splitlines(keepends=True)
    ns = {"json": json, "np": np, "Path": Path, "List": list, "Dict": dict, "tqdm": tqdm}
    exec(compile(_extract(lines, 15, 80), "save_embeddings_to_disk.py:15-80", "exec"), ns)
    save_fn = ns["save_embeddings_disk"]
    nsl = {"json": json, "np": np, "Path": Path, "List": list, "Dict": dict}
    exec(compile(_extract(GEN.read_text().splitlines(keepends=True), 76, 92), "GEN:76-92", "exec"), nsl)
    fx = GOLD / "harness"
    chunks = []
    for f in sorted(f for f in (fx / "input").rglob("*.json") if not f.name.startswith("._")):
        chunks.extend(nsl["load_chunks_from_file"](f, min_quality=0.9))
    embs = list(np.load(fx / "input_embeddings_f32.npy"))
    out = fx / "expected_batched"
    out.mkdir(exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            save_fn(chunks, embs, output_dir=td, batch_size=2)
        for f in sorted(Path(td).iterdir()):
            (out / f.name).write_bytes(f.read_bytes())
    print("batched-layout fixture:", sorted(f.name for f in out.iterdir()))


def gen_semantic():
    """Run the reference's own sentence split (:1275-1276) and grouping walk (:1542-1599, cosine helper :1601-1605)
    on synthetic sentences + embeddings; the encode in between is the GPU path and is not part of this fixture."""
    tp = GEN.parents[2] / "3-chunks/pipeline/src/processors/text_processor.py"
    if not tp.exists():
        print("reference absent; skipping semantic fixtures"); return
    lines = tp.read_text().splitlines(keepends=True)
    src = ("import re\nclass R:\n"
           "    def __init__(self, lo, hi):\n        self.min_chunk_size, self.max_chunk_size = lo, hi\n"
           "    def split(self, text):\n" + _extract(lines, 1275, 1276) + "        return sentences\n"
           "    def group(self, sentences, embeddings, metadata=None):\n" + _extract(lines, 1542, 1599)
           + _extract(lines, 1601, 1605))
    ns = {}
    exec(compile(src, "text_processor.py:1275-1276,1542-1605", "exec"), ns)
    R = ns["R"]
    rs = np.random.RandomState(11)
    words = ["graph", "neural", "network", "quantum", "field", "theory", "we", "show", "that", "the", "results",
             "lattice", "galaxy", "cluster", "proof", "lemma", "of", "a", "in", "bounded", "operator", "spectrum"]
    cases = []
    for ci, (n, lo, hi, rho) in enumerate([(40, 100, 1000, 0.9), (40, 200, 2000, 0.6), (25, 10, 300, 0.8),
                                            (60, 100, 400, 0.95), (2, 5, 1000, 0.5), (30, 100, 120, 0.9),
                                            (12, 1, 100000, 0.75)]):
        sents = []
        for _ in range(n):
            k = int(rs.randint(3, 40))
            sents.append(" ".join(words[j] for j in rs.randint(0, len(words), k)).capitalize() + rs.choice([".", "!", "?"]))
        d = 16
        e = np.zeros((n, d), np.float32)
        e[0] = rs.standard_normal(d)
        for i in range(1, n):                     # AR(1) walk: neighbouring cosines scatter around rho
            r = rho if rs.rand() > 0.25 else rs.uniform(-0.2, 0.5)
            e[i] = r * e[i - 1] / np.linalg.norm(e[i - 1]) + np.sqrt(max(1 - r * r, 0)) * rs.standard_normal(d) / np.sqrt(d)
            e[i] *= rs.uniform(0.5, 4.0)
        sims = [float(R._cosine_similarity(e[i], e[i - 1])) for i in range(1, n)]
        if any(abs(x - 0.7) < 1e-3 for x in sims):
            raise SystemExit("similarity too close to the threshold; change the seed")
        md = {"paper_id": f"0704.{ci:04d}", "section": "Introduction"} if ci % 2 == 0 else None
        chunks = R(lo, hi).group(sents, e, md)
        cases.append({"min_chunk_size": lo, "max_chunk_size": hi, "metadata": md, "sentences": sents,
                      "embeddings": e.tolist(), "similarities": sims, "expected_chunks": chunks})
    text = ("Short one. This sentence is long enough to be kept!  And so is this other one?\nTiny. "
            "A final sentence without a terminator that is kept as well")
    out = {"cases": cases, "split": {"text": text, "expected": R(0, 0).split(text)}}
    (GOLD / "semantic_chunker.json").write_text(json.dumps(out, ensure_ascii=False), encoding="utf-8")
    print("semantic fixtures:", [len(c["expected_chunks"]) for c in cases], "chunks per case; split ->", out["split"]["expected"])


if __name__ == "__main__":
    torch.manual_seed(0)
    GOLD.mkdir(parents=True, exist_ok=True)
    what = sys.argv[1:] or ["tiny", "tables", "search", "harness", "semantic", "full"]
    for w in what:
        {"tiny": gen_tiny, "full": gen_full, "tables": gen_tables, "search": gen_search,
         "harness": gen_harness, "adversarial": gen_adversarial, "semantic": gen_semantic, "batched": gen_harness_batched}[w]()
