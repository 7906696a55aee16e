#!/usr/bin/env python3
"""Dev probe: what ARX_TOPK_I8_CENTRE_QUERY costs pass A (one more fma per accumulator value in the epilogue + two more staged vectors):
int8 pass-A time on 10 M x 768 iid rows with the flag forced on and off, alternating, library events."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd import _lib
from arxiv_rag_amd.index import ShardIndex, fill_unit_rows
corpus = fill_unit_rows(10_000_000, 768, seed=7); Q = fill_unit_rows(4096, 768, seed=11)
idx = {cq: ShardIndex(corpus, prefilter="int8", centre_query=cq) for cq in (False, True)}
res = {}
for rep in range(3):
    for qb in (1, 64, 128, 256, 1024):
        for cq in (False, True):
            ix = idx[cq]
            for _ in range(2): ix.search(Q[:qb], 10)
            _lib.prof_reset(); _lib.prof_classes(None); _lib.prof_enable(True)
            for r in range(8): out = ix.search(Q[r * 64:r * 64 + qb], 10)
            torch.cuda.synchronize(); _lib.prof_enable(False)
            p = _lib.prof_read()
            k = (qb, cq)
            res[k] = min(res.get(k, 9e9), p["search_groupmax"][0] / p["search_groupmax"][1])
for qb in (1, 64, 128, 256, 1024):
    a, b = res[(qb, False)], res[(qb, True)]
    print(json.dumps({"Qb": qb, "passA_ms_query_not_centred": round(a, 4), "passA_ms_query_centred": round(b, 4), "cost": f"{(b / a - 1) * 100:+.1f} %"}))
