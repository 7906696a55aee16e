#!/usr/bin/env python3
"""Calibration probe (not product): run the vendor bf16 GEMM on the four encoder shapes so that a
`rocprofv3 --kernel-trace --stats` of this script names the library kernels (their tile configuration is in the name)."""
import torch
dev = torch.device("cuda:0")
M = 262144
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    A = torch.randn((M, K), device=dev).to(torch.bfloat16)
    W = (torch.randn((N, K), device=dev) * 0.03).to(torch.bfloat16)
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    for _ in range(6):
        torch.matmul(A, W.t(), out=C)
    torch.cuda.synchronize()
