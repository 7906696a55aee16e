#!/usr/bin/env python3
"""Same-process A/B of the attention kernels (ARX_ATTN_VARIANT is read when a handle is created): the fused attention block
alone on a bench-shaped batch (B x S tokens, mpnet-base heads), interleaved rounds, hipEvent times on the launch stream.
  python tools/attn_bench.py [variants ...]      e.g.  1 2"""
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arxiv_rag_amd import _lib, config as C
from arxiv_rag_amd.encoder import HipEncoder
from arxiv_rag_amd.weights import seeded_state_dict

variants = sys.argv[1:] or ["1", "2"]
B, S = int(os.environ.get("AB_B", 1024)), int(os.environ.get("AB_S", 256))
cfg = C.PRESETS[os.environ.get("AB_MODEL", "all-mpnet-base-v2")]
sd = seeded_state_dict(cfg, seed=0)
lib = _lib.load()
H = cfg.hidden
g = torch.Generator(device="cuda"); g.manual_seed(1)
qkv = (torch.randn((B * S, 3 * H), device="cuda", generator=g) * 1.2).to(torch.bfloat16)
ragged = os.environ.get("AB_RAGGED") == "1"
lens_h = np.full(B, S, np.int32) if not ragged else np.sort(np.random.RandomState(0).randint(S // 4, S + 1, size=B).astype(np.int32))[::-1].copy()
lens_h[0] = S
lens = torch.from_numpy(lens_h).cuda()
encs = {}
for v in variants:
    os.environ["ARX_ATTN_VARIANT"] = v
    encs[v] = HipEncoder(cfg, sd, max_tokens=B * S, max_seqs=B)
ctx = {v: torch.empty((B * S, H), dtype=torch.bfloat16, device="cuda") for v in variants}
st = torch.cuda.current_stream().cuda_stream
def run(v):
    _lib.check(lib.arx_encoder_attention(encs[v]._handle, qkv.data_ptr(), lens.data_ptr(), B, S, ctx[v].data_ptr(), st), "attn")
for v in variants:
    for _ in range(3): run(v)
torch.cuda.synchronize()
times = {v: [] for v in variants}
for r in range(12):
    for v in variants:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run(v)
        b.record(); b.synchronize()
        times[v].append(a.elapsed_time(b) / 5)
T = int(lens_h.sum())
flops = 4.0 * float((lens_h.astype(np.float64) ** 2).sum()) * H
out = {"B": B, "S": S, "tokens": T, "ragged": ragged}
for v in variants:
    t = np.array(times[v])
    out[v] = {"ms_median": round(float(np.median(t)), 4), "ms_min": round(float(t.min()), 4), "tflops": round(flops / np.median(t) / 1e9, 1),
              "hbm_GBps_algorithmic": round(T * H * 2 * 4 / np.median(t) / 1e6, 1)}
ref = ctx[variants[0]].float()
for v in variants[1:]:
    out[v]["max_abs_diff_vs_first"] = float((ctx[v].float() - ref).abs().max())
print(json.dumps(out))
