#!/usr/bin/env python3
"""Dev probe: the centred int8 index on rows with a common component (candidate counts, mu against torch's mean)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd.index import ShardIndex
d, n = 768, 64 * 3000 + 17
g = torch.Generator(device="cuda"); g.manual_seed(77)
u = torch.randn(d, generator=g, device="cuda"); u /= u.norm()
F = torch.nn.functional
for amp in (0.0, 0.15, 0.3, 0.5, 1.0):
    def rows(m):
        return F.normalize(amp * u[None, :] + 0.3 * F.normalize(torch.randn((m, d), generator=g, device="cuda"), dim=1), dim=1).half()
    C_ = rows(n); Q_ = rows(300)
    i8 = ShardIndex(C_, prefilter="int8")
    ref = C_[(torch.arange(16384, device="cuda") * n // 16384)].float().mean(0)
    print("amp", amp, "|mean|", float(ref.norm()), "cos rows", float((C_[:256].float() @ C_[256:512].float().T).mean()))
    for nq in (1, 64, 300):
        s, i = i8.search(Q_[:nq], 10)
        fl, pairs = i8.certificate_stats()
        s2, i2 = ShardIndex(C_).search(Q_[:nq], 10)
        # spread of true scores and k-th gap
        full = Q_[:nq].float() @ C_.float().T
        top = full.topk(11, dim=1).values
        print("  nq", nq, "flagged", fl, "pairs/query", pairs / nq, "same ids", float((i == i2).all(dim=1).float().mean()),
              "score std", float(full.std(dim=1).mean()), "top1-top11", float((top[:, 0] - top[:, 10]).mean()), "s10 - mean", float((top[:, 9] - full.mean(dim=1)).mean()))
