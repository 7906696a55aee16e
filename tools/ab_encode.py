#!/usr/bin/env python3
"""Dev A/B: encode step time (1024 x 256 tokens, mpnet-base shape) per library build, builds interleaved in child processes on one box.
usage: ab_encode.py lib1.so lib2.so ... [--rounds N]"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
import torch
from arxiv_rag_amd import _lib, config as C
from arxiv_rag_amd.encoder import HipEncoder
from arxiv_rag_amd.weights import seeded_state_dict
cfg = C.PRESETS["all-mpnet-base-v2"]; B, S = 1024, 256
enc = HipEncoder(cfg, seeded_state_dict(cfg, seed=0), device="cuda:0", max_tokens=B * S, max_seqs=B)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
ids = torch.randint(4, cfg.vocab_size - 1, (8, B, S), generator=g, device="cuda", dtype=torch.int32); ids[:, :, 0] = 0; ids[:, :, S - 1] = 2
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
out = torch.empty((B, cfg.hidden), dtype=torch.float16, device="cuda")
for i in range(4): enc.forward_tokens(ids[i], lens, S, B * S, out=None, out_f16=out, normalize=True)
torch.cuda.synchronize()
_lib.prof_reset(); _lib.prof_classes(None); _lib.prof_enable(True)
t0 = time.perf_counter()
K = 20
for i in range(K): enc.forward_tokens(ids[i %% 8], lens, S, B * S, out=None, out_f16=out, normalize=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
_lib.prof_enable(False)
p = _lib.prof_read()
print(json.dumps({"chunks_per_s": round(K * B / dt, 1), **{k: round(v[0] / v[1], 4) for k, v in p.items() if v[1]}}))
''' % str(ROOT)
if __name__ != "__main__":
    sys.argv = sys.argv[:1]
libs = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, ARX_LIB=str(Path(lib).resolve()))
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(Path(lib).name, o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-500:], flush=True)
