#!/usr/bin/env python3
"""Dev: raw GEMM outputs (ragged and full shapes, the three public epilogue modes, default schedule) saved to a file so that two library
builds can be compared bit for bit.  usage: ARX_LIB=... gemm_bits.py out.npz ; gemm_bits.py --cmp a.npz b.npz"""
import sys
from pathlib import Path
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        x, y = a[k], b[k]
        if not np.array_equal(x, y):
            d = np.argwhere(x != y); bad += 1
            print(k, "DIFFERS:", len(d), "elements; rows", sorted(set(d[:, 0].tolist()))[:12], "cols", sorted(set(d[:, 1].tolist()))[:12])
        else:
            print(k, "equal")
    sys.exit(1 if bad else 0)
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(5)
out = {}
for (M, N, K) in ((777, 768, 768), (300, 2304, 768), (1000, 384, 384), (517, 3072, 768), (2048 + 37, 768, 3072), (256 * 9 + 1, 1152, 384)):
    A = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn((N, K), device=dev, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn((N,), device=dev, generator=g)
    R = torch.randn((M, N), device=dev, generator=g).to(torch.bfloat16)
    for mode in (0, 1, 2):
        for v in (8, 9):
            C = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
            _lib.check(lib.arx_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, mode, v, st), "gemm")
            torch.cuda.synchronize()
            out[f"M{M}_N{N}_K{K}_mode{mode}_v{v}"] = C.view(torch.int16).cpu().numpy()
np.savez(sys.argv[1], **out)
