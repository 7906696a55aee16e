#!/usr/bin/env python3
"""Dev tool (library built with ARX_HIPCC_EXTRA=-DARX_DEV_VARIANTS): where does attention_tr_kernel's time go?

(1) per-block wall-clock stamps (block start, barrier passed = staging done, the slowest wave's end) and the hardware id of the CU a
    block ran on: staging / compute time per block, how much of a CU's time has a block computing, both resident blocks staging, ...
(2) the launch timed with probes: staging only (return at the barrier), compute only (no K/V stream), the exact running-maximum tile
    loop instead of the optimistic one — and the output difference between the two loops.
  python tools/attn_probe.py  ->  JSON lines on stdout"""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arxiv_rag_amd import _lib, config as C
from arxiv_rag_amd.encoder import HipEncoder
from arxiv_rag_amd.weights import seeded_state_dict

B, S = int(os.environ.get("AB_B", 1024)), int(os.environ.get("AB_S", 256))
cfg = C.PRESETS[os.environ.get("AB_MODEL", "all-mpnet-base-v2")]
lib = _lib.load()
assert lib.arx_build_info() & 1, "needs the dev build"
cdll = ctypes.CDLL(str(_lib.LIB_PATH))
setf = cdll.arx_dev_attn_set
setf.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]; setf.restype = ctypes.c_int32
H = cfg.hidden
g = torch.Generator(device="cuda"); g.manual_seed(1)
qkv = (torch.randn((B * S, 3 * H), device="cuda", generator=g) * 1.2).to(torch.bfloat16)
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
enc = HipEncoder(cfg, seeded_state_dict(cfg, seed=0), max_tokens=B * S, max_seqs=B)
ctx = torch.empty((B * S, H), dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
nblk = B * cfg.heads * ((S + 255) // 256)
stamps = torch.zeros((nblk, 4), dtype=torch.int64, device="cuda")
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
first = 2 * n_cu
stamps_note = "wall_clock64 ticks of 10 ns"

def run():
    _lib.check(lib.arx_encoder_attention(enc._handle, qkv.data_ptr(), lens.data_ptr(), B, S, ctx.data_ptr(), st), "attn")

def timed(word, reps=8, inner=5):
    setf(enc._handle, word, None)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner): run()
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    return float(np.median(ts)), float(min(ts))

def stamped(word):
    setf(enc._handle, word, stamps.data_ptr())
    stamps.zero_()
    run(); torch.cuda.synchronize()
    setf(enc._handle, word, None)
    s = stamps.cpu().numpy().astype(np.int64)
    t0 = s[:, 0].min()
    start, bar, end, hw = s[:, 0] - t0, s[:, 1] - t0, s[:, 2] - t0, s[:, 3]
    xcc = (hw >> 32) & 0xf
    hwid = hw & 0xffffffff
    cu = (xcc << 8) | ((hwid >> 8) & 0xff)             # xcc | se/sh/cu bits 15:8 of HW_ID
    tick = 10.0                                          # ns per wall_clock64 tick (100 MHz)
    out = {"word": word, "blocks": int(nblk), "distinct_cu": int(len(np.unique(cu))), "span_us": float(end.max() * tick / 1e3),
           "stage_us_mean": float((bar - start).mean() * tick / 1e3), "compute_us_mean": float((end - bar).mean() * tick / 1e3),
           "stage_us_p10_p90": [float(np.percentile(bar - start, 10) * tick / 1e3), float(np.percentile(bar - start, 90) * tick / 1e3)],
           "compute_us_p10_p90": [float(np.percentile(end - bar, 10) * tick / 1e3), float(np.percentile(end - bar, 90) * tick / 1e3)]}
    # per CU: fraction of the kernel's span in which (a) at least one resident block computes, (b) both resident blocks stage at once
    ov_c, ov_s, both_c = [], [], []
    T = int(end.max()) + 1
    for c in np.unique(cu)[:64]:
        m = cu == c
        comp = np.zeros(T, np.int8); stg = np.zeros(T, np.int8)
        for a, b_, e in zip(start[m], bar[m], end[m]):
            stg[a:b_] += 1; comp[b_:e] += 1
        ov_c.append((comp > 0).mean()); ov_s.append((stg > 1).mean()); both_c.append((comp > 1).mean())
    out.update({"cu_frac_some_block_computing": float(np.mean(ov_c)), "cu_frac_two_blocks_staging": float(np.mean(ov_s)),
                "cu_frac_two_blocks_computing": float(np.mean(both_c))})
    # where did the first-round blocks land: linear id -> CU, for the slot mapping
    out["same_cu_lin_and_lin_plus_ncu"] = float((cu[:n_cu] == cu[n_cu:2 * n_cu]).mean())
    out["same_cu_lin_even_odd"] = float((cu[0:first:2] == cu[1:first:2]).mean())
    return out

print(json.dumps({"n_cu": n_cu, "B": B, "S": S, "lib": str(_lib.LIB_PATH)}), flush=True)
res = {}
if "--quick" in sys.argv:                      # timings only (bisect builds: -DARX_ATTN_PROBE=k give wrong results by design)
    for rnd in range(3):
        for name, w in (("product", 0), ("compute_only", 5)):
            res.setdefault(name, []).append(round(timed(w, reps=6)[0], 4))
    print(json.dumps({"lib": os.path.basename(str(_lib.LIB_PATH)), "ms_median": res}), flush=True)
    sys.exit(0)
for rnd in range(2):
    for name, w in (("product", 0), ("exact_loop", 6), ("stage_only", 3), ("compute_only", 5)):
        res.setdefault(name, []).append(timed(w))
print(json.dumps({"timed_ms_median_min": res}), flush=True)
setf(enc._handle, 0, None); run(); torch.cuda.synchronize(); a = ctx.float().clone()
setf(enc._handle, 6, None); run(); torch.cuda.synchronize(); b = ctx.float().clone()
setf(enc._handle, 0, None)
print(json.dumps({"optimistic_vs_exact": {"max_abs_diff": float((a - b).abs().max()), "frac_differ": float((a != b).float().mean()),
                                          "max_abs_v": float(qkv[:, 2 * H:].float().abs().max())}}), flush=True)
print(json.dumps({"stamps_product": stamped(0)}), flush=True)
print(json.dumps({"stamps_exact_loop": stamped(6)}), flush=True)
