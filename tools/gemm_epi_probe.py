#!/usr/bin/env python3
"""Dev probe (library built with -DARX_DEV_VARIANTS [-DARX_STAMP], ARX_LIB pointing at it): what the persistent GEMM's epilogue
costs and whether the CUs of the chip go through it in lockstep.
  * same launch with the epilogue's stores skipped (variant + 300; a "residual loads skipped" arm existed while section 7 of
    profiles/r03/gemm_epilogue_store_probe.md was measured),
  * blocks started out of phase (ARX_DEV_STAGGER=cycles,slots: block b waits (b / 8 % slots) * cycles),
  * with ARX_STAMP: raw per-tile stamps to gpurun_out/stamps_<shape>_<cfg>.bin ([tiles][2 groups][4] u64).
usage: gemm_epi_probe.py [outdir]"""
import os, sys, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from arxiv_rag_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
out = Path(sys.argv[1] if len(sys.argv) > 1 else ROOT / "gpurun_out" / "epi_probe"); out.mkdir(parents=True, exist_ok=True)
st = torch.cuda.current_stream().cuda_stream
M = 262144
shapes = [("qkv", 2304, 768, 0), ("oproj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)]
if os.environ.get("EPI_PROBE_SHAPES"): shapes = [s for s in shapes if s[0] in os.environ["EPI_PROBE_SHAPES"].split(",")]
cfgs = [("base", 9, None), ("nostore", 309, None), ("stg5500x8", 9, "5500,8"), ("pertile", 8, None), ("pertile_nostore", 308, None)]
stamped = bool(lib.arx_build_info() & 2) if hasattr(lib, "arx_build_info") else False
g = torch.Generator(device=dev); g.manual_seed(0)

def run(v, stg, A, W, b, R, C, mode):
    if stg: os.environ["ARX_DEV_STAGGER"] = stg
    else: os.environ.pop("ARX_DEV_STAGGER", None)
    _lib.check(lib.arx_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr() if R is not None else None, C.data_ptr(),
                                 A.shape[0], W.shape[0], A.shape[1], mode, v, st), "gemm")

for name, N, K, mode in shapes:
    A = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn((N, K), device=dev, generator=g) * 0.03).to(torch.bfloat16)
    b = torch.randn((N,), device=dev, generator=g) * 0.1
    R = torch.randn((M, N), device=dev, generator=g).to(torch.bfloat16) if mode == 2 else None
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    t = {c[0]: [] for c in cfgs}
    for c in cfgs: run(c[1], c[2], A, W, b, R, C, mode)
    torch.cuda.synchronize()
    for rnd in range(5):
        for cn, v, stg in cfgs:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run(v, stg, A, W, b, R, C, mode)
            e1.record(); e1.synchronize()
            t[cn].append(e0.elapsed_time(e1) / 5)
    flops = 2.0 * M * N * K
    row = {"shape": name, **{cn: round(float(np.median(v)), 4) for cn, v in t.items()},
           "TF_base": round(flops / np.median(t["base"]) / 1e9, 1)}
    print(json.dumps(row), flush=True)
    with open(out / "timing.jsonl", "a") as f: f.write(json.dumps(row) + "\n")
    if os.environ.get("EPI_PROBE_STAMPS") == "1":
        os.environ["ARX_STAMP_DUMP"] = "1"
        for cn, v, stg in (("base", 9, None), ("nostore", 309, None), ("pertile", 8, None)):
            os.environ["ARX_STAMP_FILE"] = str(out / f"stamps_{name}_{cn}.bin")
            run(v, stg, A, W, b, R, C, mode)
            torch.cuda.synchronize()
        os.environ.pop("ARX_STAMP_DUMP"); os.environ.pop("ARX_STAMP_FILE")
