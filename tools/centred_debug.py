#!/usr/bin/env python3
"""Dev probe: the centred int8 index on rows with a common component — candidate (query, group) pairs per query and overflowing queries,
with the query centred as well (ARX_TOPK_I8_CENTRE_QUERY) and without; answers compared with the fp16 pass."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd.index import ShardIndex
d, n = 768, 64 * 3000 + 17
g = torch.Generator(device="cuda"); g.manual_seed(77)
u = torch.randn(d, generator=g, device="cuda"); u /= u.norm()
F = torch.nn.functional
for amp in (0.0, 0.15, 0.3, 0.5, 1.0, 2.0):
    def rows(m):
        return F.normalize(amp * u[None, :] + 0.3 * F.normalize(torch.randn((m, d), generator=g, device="cuda"), dim=1), dim=1).half()
    C_ = rows(n); Q_ = rows(300)
    ref = ShardIndex(C_)
    print("amp", amp, "mean cosine between rows", round(float((C_[:256].float() @ C_[256:512].float().T).mean()), 4), flush=True)
    for cq in (False, True):
        i8 = ShardIndex(C_, prefilter="int8", centre_query=cq)
        out = []
        for nq in (1, 64, 300):
            s, i = i8.search(Q_[:nq], 10)
            fl, pairs = i8.certificate_stats()
            s2, i2 = ref.search(Q_[:nq], 10)
            out.append(f"nq {nq}: pairs/query {pairs / nq:.0f} overflowed {fl} same ids {float((i == i2).all(dim=1).float().mean()):.3f}")
        print(f"   centre_query={cq} |mu|={i8.i8_mean_norm:.3f}  " + " | ".join(out), flush=True)
