#!/usr/bin/env python3
"""One query-sized batch through the small-batch schedule, many times: the program to put under `rocprofv3 --kernel-trace --stats`
(per-kernel time of a query forward).  python tools/query_profile.py [nq] [L] [low_latency 0|1]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arxiv_rag_amd import config as C
from arxiv_rag_amd.encoder import HipEncoder
from arxiv_rag_amd.weights import seeded_state_dict

nq, L, ll = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 12, (sys.argv[3] != "0") if len(sys.argv) > 3 else True
cfg = C.PRESETS[os.environ.get("QP_MODEL", "all-mpnet-base-v2")]
enc = HipEncoder(cfg, seeded_state_dict(cfg, seed=0))
rs = np.random.RandomState(0)
ids = rs.randint(4, cfg.vocab_size - 1, size=(nq, L)).astype(np.int32); ids[:, 0] = 0; ids[:, -1] = 2
d_ids = torch.from_numpy(ids).cuda(); d_lens = torch.full((nq,), L, dtype=torch.int32, device="cuda")
o = torch.empty((nq, cfg.hidden), dtype=torch.float32, device="cuda")
for _ in range(200):
    enc.forward_tokens(d_ids, d_lens, L, nq * L, out=o, low_latency=ll)
torch.cuda.synchronize()
