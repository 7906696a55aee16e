#!/usr/bin/env python3
"""Dev probe: run one GEMM shape/variant in a loop for a few seconds while sampling `rocm-smi` clocks and power
(is the kernel power-limited?).  usage: [POWER_PROBE_ZERO=1] power_probe.py <variant> [shape]"""
import subprocess, sys, threading, time, re
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
v = int(sys.argv[1]); shape = sys.argv[2] if len(sys.argv) > 2 else "qkv"
N, K, mode = {"qkv": (2304, 768, 0), "fc1": (3072, 768, 1), "fc2": (768, 3072, 2)}[shape]
M = 262144
A = torch.randn((M, K), device=dev).to(torch.bfloat16); W = (torch.randn((N, K), device=dev) * 0.03).to(torch.bfloat16)
import os
if os.environ.get("POWER_PROBE_ZERO") == "1":      # zero operands: the same instruction stream with (almost) no data toggling
    A.zero_(); W.zero_()
b = torch.randn((N,), device=dev); R = torch.randn((M, N), device=dev).to(torch.bfloat16); C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
samples = []; stop = False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out); pw = re.findall(r"Power \(W\): ([\d.]+)", out)
        samples.append((sclk[:1], pw[:1]))
        time.sleep(0.2)
def run(n):
    for _ in range(n):
        lib.arx_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, mode, v, st)
run(20); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); run(4000); e1.record(); torch.cuda.synchronize()
stop = True; th.join()
ms = e0.elapsed_time(e1) / 4000
print(f"variant {v} {shape}: {ms:.4f} ms/launch, {2.0*M*N*K/ms/1e9:.1f} TF over {time.time()-t0:.1f} s")
print("samples (sclk MHz, W):", samples[2:-1][:12])
