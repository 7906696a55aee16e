#!/usr/bin/env python3
"""Dev: where does the aux epilogue's 4 % of pass A go?  pass-A time (library events) on the 625 k x 768 shard at 64 / 256 queries per build:
the shipped library, ARX_AUXDBG=1 (aux computed, NOT stored), ARX_AUXDBG=2 (no keyed maxima, a constant stored), and the shipped library with
the pair path (no aux at all); builds alternate in child processes.
(The two timing builds were one-off edits of groupmax_epilogue_f16_aux behind -DARX_AUXDBG — skip the second store_query_row / replace the keyed
maxima by a constant — and are not in the tree; result: profiles/r04/aux_epilogue_cost_probe.txt.)"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import torch
from arxiv_rag_amd import _lib
from arxiv_rag_amd.index import ShardIndex, fill_unit_rows
corpus = fill_unit_rows(625_000, 768, seed=7); Q = fill_unit_rows(4096, 768, seed=11)
idx = ShardIndex(corpus)
out = {}
for qb in (64, 256):
    for tag, fl in (("aux", 0), ("pair", _lib.TOPK_NO_SINGLE_ROW_TAIL)):
        o_scan = idx._options(flags=_lib.TOPK_SCAN_ONLY | fl)
        ws = idx.alloc_workspace(qb, 10)
        so = (torch.empty((qb, 10), dtype=torch.float32, device="cuda"), torch.empty((qb, 10), dtype=torch.int64, device="cuda"))
        for _ in range(5): idx.search(Q[:qb], 10, ws=ws, out=so, _opt=o_scan)
        _lib.prof_reset(); _lib.prof_classes(["search_groupmax"]); _lib.prof_enable(True)
        for r in range(40): idx.search(Q[r * 7:r * 7 + qb], 10, ws=ws, out=so, _opt=o_scan)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        p = _lib.prof_read()["search_groupmax"]
        out[f"Qb={qb} {tag}"] = round(p[0] / p[1], 4)
print(json.dumps(out))
''' % str(ROOT)
libs = ["libarx_hip.so", "libarx_aux1.so", "libarx_aux2.so"]
for r in range(2):
    for lib in libs:
        env = dict(os.environ, ARX_LIB=str(ROOT / "arxiv_rag_amd" / lib))
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(lib, o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-400:], flush=True)
