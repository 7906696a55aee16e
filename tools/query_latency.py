#!/usr/bin/env python3
"""Single-query latency of the encode step (the front half of QPS at Qb = 1): eager launches vs one captured HIP graph.

  python tools/query_latency.py [model]   ->  JSON on stdout
Four ways per batch shape: the default schedule (256 x 256 GEMM tiles) and the small-batch schedule (`low_latency=True`: csrc/gemm_small.h,
for <= 256 token rows), each as eager launches and as one captured HIP graph (torch.cuda.CUDAGraph around the same call, static buffers)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arxiv_rag_amd import config as C
from arxiv_rag_amd.encoder import HipEncoder
from arxiv_rag_amd.weights import seeded_state_dict

name = sys.argv[1] if len(sys.argv) > 1 else "all-mpnet-base-v2"
cfg = C.PRESETS[name]
enc = HipEncoder(cfg, seeded_state_dict(cfg, seed=0))
rs = np.random.RandomState(0)
out = {"model": name}
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 4)


def captured(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    return g


for (nq, L) in ((1, 12), (4, 16), (16, 16), (16, 24), (64, 16), (64, 32), (256, 32), (512, 32)):
    ids = rs.randint(4, cfg.vocab_size - 1, size=(nq, L)).astype(np.int32); ids[:, 0] = 0; ids[:, -1] = 2
    lens = np.full(nq, L, np.int32)
    d_ids = torch.from_numpy(ids).cuda(); d_lens = torch.from_numpy(lens).cuda()
    o = torch.empty((nq, cfg.hidden), dtype=torch.float32, device="cuda")
    res = {}
    for ll in (False, True):
        fn = lambda: enc.forward_tokens(d_ids, d_lens, L, nq * L, out=o, low_latency=ll)
        tag = "small_batch" if ll else "default"
        res[tag + "_eager_ms"] = timed(fn)
        ref = o.clone()
        g = captured(fn)
        o.zero_(); g.replay(); torch.cuda.synchronize()
        res[tag + "_graph_ms"] = timed(g.replay)
        res[tag + "_graph_bit_identical"] = bool(torch.equal(o, ref))
        if ll:
            res["cos_min_vs_default"] = float(torch.nn.functional.cosine_similarity(o, ref_default, dim=1).min())
        else:
            ref_default = ref
    res["schedule"] = "split-K wave tiles" if nq * L <= 256 else ("128 x 128 tiles" if nq * L <= 8192 else "default")
    out[f"nq={nq},L={L}"] = res
print(json.dumps(out))
