#!/usr/bin/env python3
"""Dev: the search kernels at one (rows, Qb) under rocprofv3 --kernel-trace --stats.  usage: i8_tail_profile.py <rows> <qb> [flags] [fp16|int8]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd.index import ShardIndex, fill_unit_rows
N, qb = int(sys.argv[1]), int(sys.argv[2]); fl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pre = None if (len(sys.argv) > 4 and sys.argv[4] == "fp16") else "int8"
corpus = fill_unit_rows(N, 768, seed=7); Q = fill_unit_rows(4096, 768, seed=11)
idx = ShardIndex(corpus, prefilter=pre)
for r in range(24):
    idx.search(Q[r * 7:r * 7 + qb], 10, flags=fl)
torch.cuda.synchronize()
