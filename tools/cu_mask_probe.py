#!/usr/bin/env python3
"""Dev probe: (1) which CUs a CU-masked stream really uses (arx_debug_cu_census), per XCD; (2) the pipelined 625 k-row search
(ShardIndex.search_many) for several scan / tail CU splits, against batch-by-batch `search`."""
import collections, json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import ctypes as C
import torch
from arxiv_rag_amd import _lib
from arxiv_rag_amd.index import ShardIndex, fill_unit_rows

lib = _lib.load(); dev = torch.device("cuda:0")
n_cu = lib.arx_device_cu_count()
print("CUs", n_cu)


def census(stream_ptr, n_blocks=8192, spin=200000):
    out = torch.zeros(n_blocks, dtype=torch.int32, device=dev)
    _lib.check(lib.arx_debug_cu_census(out.data_ptr(), n_blocks, spin, stream_ptr), "census")
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype("uint32")
    per_xcc = collections.defaultdict(set)
    for x in v:
        per_xcc[int(x >> 16)].add(int(x & 0xffff) >> 8)            # (SE, SH, CU) bits
    return {k: len(s) for k, s in sorted(per_xcc.items())}


print("default stream:", census(torch.cuda.current_stream().cuda_stream))
words = (n_cu + 31) // 32
for lo, hi in ((0, 224), (224, 256), (0, 8), (0, 32), (128, 256)):
    m = (C.c_uint32 * words)()
    for b in range(lo, hi):
        m[b // 32] |= 1 << (b % 32)
    h = C.c_void_p(None)
    _lib.check(lib.arx_stream_create_cu_mask(m, words, C.byref(h)), "mask")
    print(f"mask bits [{lo},{hi}):", census(h.value))
    lib.arx_stream_destroy(h)

N, D = int(os.environ.get("PROBE_ROWS", 625_000)), 768
corpus = fill_unit_rows(N, D, seed=7); Q = fill_unit_rows(4096, D, seed=11)
for pre in (None, "int8"):
    idx = ShardIndex(corpus, prefilter=pre)
    for qb in (64, 256):
        nb = 32
        batches = [Q[(r * qb) % (4096 - qb):(r * qb) % (4096 - qb) + qb] for r in range(nb)]
        for b in batches[:3]: idx.search(b, 10)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in batches: idx.search(b, 10)
        torch.cuda.synchronize(); base = (time.perf_counter() - t0) / nb * 1e3
        row = {"prefilter": pre, "Qb": qb, "batch_by_batch_ms": round(base, 4), "roofline_ms": round(N * D * 2 / 8e12 * 1e3, 4)}
        for tail in (0, 16, 32, 48, 64, 96):
            idx.search_many(batches[:4], 10, tail_cus=tail)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            idx.search_many(batches, 10, tail_cus=tail)
            torch.cuda.synchronize(); row[f"tail{tail}_ms"] = round((time.perf_counter() - t0) / nb * 1e3, 4)
        print(json.dumps(row))
