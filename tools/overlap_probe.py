#!/usr/bin/env python3
"""Dev probe: the pipelined 625 k-row search (ShardIndex.search_many) over lanes x CU split, against batch-by-batch `search` and the scan
phase alone on its CU subset."""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd import _lib
from arxiv_rag_amd.index import ShardIndex, fill_unit_rows, _cu_streams

lib = _lib.load(); dev = torch.device("cuda:0")
N, D, NB = int(os.environ.get("PROBE_ROWS", 625_000)), 768, 48
corpus = fill_unit_rows(N, D, seed=7); Q = fill_unit_rows(4096, D, seed=11)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3


for pre in (None, "int8"):
    idx = ShardIndex(corpus, prefilter=pre)
    for QB in (64, 256):
        batches = [Q[(r * QB) % (4096 - QB):(r * QB) % (4096 - QB) + QB] for r in range(NB)]
        for b in batches[:3]: idx.search(b, 10)
        row = {"prefilter": pre, "Qb": QB, "roofline_ms": round(N * D * 2 / 8e12 * 1e3, 4),
               "batch_by_batch_ms": round(timed(lambda: [idx.search(b, 10) for b in batches]) / NB, 4)}
        outs = [(torch.empty((QB, 10), dtype=torch.float32, device=dev), torch.empty((QB, 10), dtype=torch.int64, device=dev)) for _ in range(2)]
        wss = [idx.alloc_workspace(QB, 10) for _ in range(2)]
        for tail_cus in (0, 16, 32, 64):
            scan, tail, scan_cus = _cu_streams(dev, tail_cus)
            o_scan = idx._options(cu_limit=scan_cus, flags=_lib.TOPK_SCAN_ONLY)
            def scans():
                for b in range(NB):
                    idx.search(batches[b], 10, ws=wss[b & 1], out=outs[b & 1], _opt=o_scan, _stream=scan.cuda_stream)
            scans()
            row[f"scan_alone_t{tail_cus}"] = round(timed(scans) / NB, 4)
            for lanes in (1, 2, 3):
                idx.search_many(batches[:6], 10, tail_cus=tail_cus, lanes=lanes)
                row[f"many_t{tail_cus}_l{lanes}"] = round(timed(lambda: idx.search_many(batches, 10, tail_cus=tail_cus, lanes=lanes)) / NB, 4)
        print(json.dumps(row), flush=True)
