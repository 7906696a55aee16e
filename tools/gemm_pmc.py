#!/usr/bin/env python3
"""Dev tool for PMC passes: each variant given on the command line runs the four encoder GEMM shapes 3x (the kernel template
arguments in the profile tell the variants apart).  usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python3 tools/gemm_pmc.py 8 80"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
M = 262144
for name, N, K, mode in (("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)):
    A = torch.randn((M, K), device=dev).to(torch.bfloat16); W = (torch.randn((N, K), device=dev) * 0.03).to(torch.bfloat16)
    b = torch.randn((N,), device=dev); R = torch.randn((M, N), device=dev).to(torch.bfloat16); C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    for v in [int(x) for x in sys.argv[1:]]:
        for _ in range(3):
            _lib.check(lib.arx_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, mode, v, st), name)
    torch.cuda.synchronize()
