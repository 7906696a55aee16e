#!/usr/bin/env python3
"""Diagnostic run on the GPU box: per-layer error of the HIP encoder vs the numpy oracle, search parity,
both GEMM staging variants.  Prints a report; exit code 1 on any hard failure."""
import os, sys, time, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from arxiv_rag_amd import config as C
from arxiv_rag_amd.weights import seeded_state_dict
from arxiv_rag_amd.encoder import HipEncoder
from arxiv_rag_amd.index import ShardIndex, merge_partials, fill_unit_rows
from oracle import encoder_oracle as EO, search_oracle as SO

GOLD = ROOT / "tests" / "golden"
fail = 0

def cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1) + 1e-30)

def tiny(name, glds):
    global fail
    os.environ["ARX_GEMM_GLDS"] = "1" if glds else "0"
    g = np.load(GOLD / f"{name}.npz")
    cfg = C.PRESETS[name]
    sd = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    ids, lens = g["ids"], g["lens"]
    enc = HipEncoder(cfg, sd)
    valid = (np.arange(ids.shape[1])[None] < lens[:, None])
    for layer in range(cfg.layers + 1):
        hid = enc.tap_hidden(ids, lens, layer)
        ref = g[f"hidden_{layer}"][valid]
        err = np.abs(hid - ref).max(); rel = err / (np.abs(ref).max() + 1e-9)
        print(f"  {name} glds={glds} layer {layer}: max abs err {err:.4f} (ref max {np.abs(ref).max():.3f}) nan={np.isnan(hid).any()}")
    emb = enc.encode_tokens(ids, lens).cpu().numpy()
    c = cos(emb, g["emb"])
    print(f"  {name} glds={glds} emb cos min {c.min():.6f}  norms {np.linalg.norm(emb,axis=1).round(4)[:4]}")
    if not (c.min() > 1 - 1e-3): fail += 1
    enc.close()

def full(name, key, rows, glds=True):
    global fail
    os.environ["ARX_GEMM_GLDS"] = "1" if glds else "0"
    g = np.load(GOLD / "full_shapes.npz")
    cfg = C.PRESETS[name]
    seed, std, bstd, jit = g[key + ":wspec"]
    t0 = time.time()
    sd = seeded_state_dict(cfg, seed=int(seed), std=std, bias_std=bstd, ln_jitter=jit)
    ids, lens, ref = g[key + ":ids"], g[key + ":lens"], g[key + ":emb"]
    enc = HipEncoder(cfg, sd)
    emb = enc.encode_tokens(ids, lens).cpu().numpy()
    c = cos(emb, ref)
    print(f"  {key} glds={glds}: cos min {c.min():.6f} mean {c.mean():.6f} ({time.time()-t0:.1f}s)", c.round(5).tolist())
    # ragged re-bucketing must not change results
    seqs = [ids[r, :lens[r]].tolist() for r in range(len(lens))]
    emb2 = enc.encode_ragged(seqs, batch_size=5)
    c2 = cos(emb2, ref)
    print(f"    ragged(batch 5) cos min {c2.min():.6f}")
    if not (c.min() > 1 - 1e-3 and c2.min() > 1 - 1e-3): fail += 1
    enc.close()

def search(glds):
    global fail
    os.environ["ARX_GEMM_GLDS"] = "1" if glds else "0"
    g = np.load(GOLD / "search_4096x768.npz")
    Cm = SO.unit_rows_f16(4096, 768, 7); Q = SO.unit_rows_f16(64, 768, 11)
    Cm[100] = Cm[17]; Cm[2000] = Cm[17]; Cm[3000] = Cm[17]; Q[0] = Cm[17]
    idx = ShardIndex(torch.from_numpy(Cm).cuda())
    s, i = idx.search(torch.from_numpy(Q).cuda(), 10)
    s, i = s.cpu().numpy(), i.cpu().numpy()
    same = (i == g["ids"]).all(axis=1)
    print(f"  search 4096x768 glds={glds}: rows identical {same.sum()}/64, score maxdiff {np.abs(s-g['scores']).max():.2e}, q0 {i[0,:5].tolist()}")
    if same.sum() < 64:
        bad = np.where(~same)[0][:3]
        for b in bad: print("    q", b, "got", i[b].tolist(), "want", g["ids"][b].tolist(), "scores", s[b].round(5).tolist())
        fail += 1
    # ragged sizes, several query-batch shapes, k > n
    for (n, nq, d, k) in ((5000, 77, 768, 10), (130, 3, 128, 10), (7, 2, 64, 10), (70000, 200, 384, 10), (3000, 300, 1024, 5), (9000, 1500, 256, 10)):
        Cm = SO.unit_rows_f16(n, d, 3); Q = SO.unit_rows_f16(nq, d, 4)
        idx = ShardIndex(torch.from_numpy(Cm).cuda(), idx_base=1000)
        s, i = idx.search(torch.from_numpy(Q).cuda(), k)
        s, i = s.cpu().numpy(), i.cpu().numpy()
        rs, ri = SO.topk_search(Cm, Q, k, idx_base=1000)
        seteq = sum(set(a.tolist()) == set(b.tolist()) for a, b in zip(i, ri))
        fin = np.isfinite(rs)
        print(f"  search n={n} nq={nq} d={d} k={k} glds={glds}: set-equal {seteq}/{nq} exact-order {(i==ri).all(axis=1).sum()}/{nq} score maxdiff {np.abs(s[fin]-rs[fin]).max():.2e}")
        if seteq < nq: fail += 1

def merge():
    global fail
    rs = np.random.RandomState(0)
    P, nq, k = 8, 37, 10
    s = rs.standard_normal((P, nq, k)).astype(np.float32); s = -np.sort(-s, axis=2)
    i = rs.randint(0, 10**9, size=(P, nq, k)).astype(np.int64)
    s[3, :, 5:] = -np.inf; i[3, :, 5:] = -1
    s[1, 0, 0] = s[2, 0, 0] = 9.0; i[1, 0, 0] = 500; i[2, 0, 0] = 100
    ms, mi = merge_partials(torch.from_numpy(s).cuda(), torch.from_numpy(i).cuda(), k)
    rs_, ri_ = SO.merge_partials(s, i, k)
    ok = (mi.cpu().numpy() == ri_).all() and np.array_equal(ms.cpu().numpy(), rs_)
    print("  merge kernel exact:", ok)
    if not ok: fail += 1

if __name__ == "__main__":
    print("device:", torch.cuda.get_device_name(0))
    which = sys.argv[1:] or ["tiny", "search", "merge", "full"]
    for glds in (False, True):
        if "tiny" in which:
            for n in ("tiny-mpnet", "tiny-bert", "tiny-bert-cls"):
                try: tiny(n, glds)
                except Exception as e: print("  EXC", n, glds, repr(e)); fail += 1
        if "search" in which:
            try: search(glds)
            except Exception as e: print("  EXC search", glds, repr(e)); fail += 1
    if "merge" in which: merge()
    if "full" in which:
        full("all-MiniLM-L6-v2", "all-MiniLM-L6-v2:w05", None)
        full("all-mpnet-base-v2", "all-mpnet-base-v2:w05", None)
        full("all-mpnet-base-v2", "all-mpnet-base-v2:hf02", None)
        full("all-mpnet-base-v2", "all-mpnet-base-v2:w05", None, glds=False)
        full("BAAI/bge-large-en-v1.5", "BAAI_bge-large-en-v1.5:w05", None)
    print("FAILURES:", fail)
    sys.exit(1 if fail else 0)
