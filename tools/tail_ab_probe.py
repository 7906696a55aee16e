#!/usr/bin/env python3
"""Dev probe, same box / same binary: small fp16 batches with the single-row tail (aux words in pass A) against the select + rescore pair
(`flags=ARX_TOPK_NO_SINGLE_ROW_TAIL`), per shard size and query-batch size; arms interleaved."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd import _lib
from arxiv_rag_amd.index import ShardIndex, fill_unit_rows

D = 768
Q = fill_unit_rows(4096, D, seed=11)
for N in (625_000, 2_000_000, 5_000_000, 10_000_000):
    corpus = fill_unit_rows(N, D, seed=7)
    idx = ShardIndex(corpus)
    for qb in (1, 64, 128, 256):
        row = {"rows": N, "Qb": qb}
        for rep in range(2):
            for tag, fl in (("single", 0), ("pair", _lib.TOPK_NO_SINGLE_ROW_TAIL)):
                for _ in range(3): idx.search(Q[:qb], 10, flags=fl)
                _lib.prof_reset(); _lib.prof_classes(None); _lib.prof_enable(True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                n = 20 if N <= 2_000_000 else 8
                for r in range(n): idx.search(Q[r * 7:r * 7 + qb], 10, flags=fl)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
                _lib.prof_enable(False)
                p = _lib.prof_read()
                row[f"{tag}{rep}"] = [round(dt, 4), round(p["search_groupmax"][0] / n, 4), round((p["search_select"][0] + p["search_rescore"][0]) / n, 4)]
        print(json.dumps(row), flush=True)
    del idx, corpus
    torch.cuda.empty_cache()
