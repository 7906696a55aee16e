#!/usr/bin/env python3
"""Report the parity margin (min / mean per-vector cosine against the golden fp32 embeddings) for every full-shape fixture,
per GEMM schedule.  usage: parity_margin.py"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import numpy as np
from arxiv_rag_amd import config as C
from arxiv_rag_amd.weights import seeded_state_dict
g = np.load(ROOT / "tests" / "golden" / "full_shapes.npz")
keys = [("all-mpnet-base-v2", "all-mpnet-base-v2:w05"), ("all-mpnet-base-v2", "all-mpnet-base-v2:hf02"),
        ("all-MiniLM-L6-v2", "all-MiniLM-L6-v2:w05"), ("all-MiniLM-L6-v2", "all-MiniLM-L6-v2:hf02"),
        ("BAAI/bge-large-en-v1.5", "BAAI_bge-large-en-v1.5:w05")]
def cos(a, b): return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
for variant in (sys.argv[1:] or ["89", "13"]):
    os.environ["ARX_GEMM_VARIANT"] = variant
    from arxiv_rag_amd.encoder import HipEncoder
    for name, key in keys:
        cfg = C.PRESETS[name]
        seed, std, bstd, jit = g[key + ":wspec"]
        sd = seeded_state_dict(cfg, seed=int(seed), std=std, bias_std=bstd, ln_jitter=jit)
        enc = HipEncoder(cfg, sd)
        emb = enc.encode_tokens(g[key + ":ids"], g[key + ":lens"]).cpu().numpy()
        c = cos(emb.astype(np.float64), g[key + ":emb"].astype(np.float64))
        print(f"variant {variant} {key:32s} 1-cos: max {1 - c.min():.2e} mean {1 - c.mean():.2e}")
        enc.close()
