#!/usr/bin/env python3
"""Dev probe: pass-A time of the int8 / fp16 search kernels at small query batches (library events), N x D shard generated in HBM."""
import sys, time, json, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd import _lib
from arxiv_rag_amd.index import ShardIndex, fill_unit_rows

N, D = int(os.environ.get("PROBE_ROWS", 10_000_000)), 768
corpus = fill_unit_rows(N, D, seed=7)
Q = fill_unit_rows(2048, D, seed=11)
out = {}
for tag, pre in (("fp16", None), ("int8", "int8")):
    idx = ShardIndex(corpus, prefilter=pre)
    for qb in (1, 16, 64, 128, 256, 1024):
        fl = _lib.TOPK_NO_PERSISTENT if os.environ.get("PROBE_NO_PERSISTENT") == "1" else 0     # A/B arm: the per-tile pass-A kernel
        for _ in range(2): idx.search(Q[:qb], 10, flags=fl)
        _lib.prof_reset(); _lib.prof_classes(None); _lib.prof_enable(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r in range(10): idx.search(Q[r * 7:r * 7 + qb], 10, flags=fl)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        _lib.prof_enable(False)
        p = _lib.prof_read()
        out[f"{tag} Qb={qb}"] = {"batch_ms": round(dt * 1e3, 4), "passA_ms": round(p["search_groupmax"][0] / p["search_groupmax"][1], 4),
                                 "select_ms": round(p["search_select"][0] / 10, 4), "rescore_ms": round(p["search_rescore"][0] / 10, 4)}
    del idx
for k, v in out.items():
    print(k, json.dumps(v))
