#!/usr/bin/env python3
"""Dev probe: pass-A time (library events, best of 2) against the number of queries of a narrow batch, int8 and fp16 rows, 10 M x 768."""
import sys, time, json
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd import _lib
from arxiv_rag_amd.index import ShardIndex, fill_unit_rows
N, D = 10_000_000, 768
corpus = fill_unit_rows(N, D, seed=7); Q = fill_unit_rows(4096, D, seed=11)
for tag, pre in (("int8", "int8"), ("fp16", None)):
    idx = ShardIndex(corpus, prefilter=pre)
    res = {}
    for rep in range(2):
        for qb in (1, 8, 16, 17, 24, 32, 33, 40, 48, 56, 64):
            for _ in range(2): idx.search(Q[:qb], 10)
            _lib.prof_reset(); _lib.prof_classes(None); _lib.prof_enable(True)
            for r in range(10): idx.search(Q[r * 64:r * 64 + qb], 10)
            torch.cuda.synchronize(); _lib.prof_enable(False)
            p = _lib.prof_read()
            pa = p["search_groupmax"][0] / p["search_groupmax"][1]
            res[qb] = min(res.get(qb, 9e9), round(pa, 4))
    print(tag, json.dumps(res), flush=True)
    del idx
