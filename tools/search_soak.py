#!/usr/bin/env python3
"""Race / edge screen of the search path (the counterpart of tools/gemm_soak.py): random shard sizes (1 row ... a few hundred thousand, ragged
against every tile size), dimensions, query-batch sizes (1 ... 1 100: narrow tiles, the persistent pass, two internal passes), k, first pass
(fp16 / int8), row norms (unit / 0.01 ... 8), a component shared by all rows (anisotropic embeddings), planted near-ties and duplicated rows, idx_base; every answer validated ON THE DEVICE against
fp32 scores of EVERY row (returned scores are those rows' scores; nothing left out beats the k-th by more than the tolerance; equal scores in
ascending id order), run twice (bit-repeatable), and — batches of <= 1 024 queries — against `search_many` (2 lanes).
usage: search_soak.py [seconds] [seed]"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from arxiv_rag_amd import _lib
from arxiv_rag_amd.index import ShardIndex

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
LOG = open(sys.argv[3], "w") if len(sys.argv) > 3 else None
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator(device="cuda"); g.manual_seed(seed)
gc = torch.Generator(); gc.manual_seed(seed)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gc).item())


def validate(c, q, s, i, k, base, tol, what):
    # [nq, n] fp32 scores of the fp16 values, in column chunks whose OUTPUT stays below 2^30 bytes: torch's matmul on this stack writes
    # zeros past element 2^29 of a larger result (found the hard way: 250 x 2.35 M "failed" from query 228 on — the reference did)
    step = max(1, (1 << 28) // max(1, q.shape[0]))
    full = torch.cat([q.float() @ c[a:a + step].float().T for a in range(0, c.shape[0], step)], dim=1)
    n = c.shape[0]
    kk = min(k, n)
    ids = i[:, :kk] - base
    assert ((ids >= 0) & (ids < n)).all(), what
    srt = ids.sort(dim=1).values
    assert (srt[:, 1:] != srt[:, :-1]).all(), (what, "duplicate ids")
    ref = full.gather(1, ids)
    assert (ref - s[:, :kk]).abs().max().item() <= 1e-5 * max(1.0, tol / 2e-6), (what, "scores", (ref - s[:, :kk]).abs().max().item())
    masked = full.clone(); masked.scatter_(1, ids, float("-inf"))
    worst = (masked.max(dim=1).values - ref.min(dim=1).values).max().item() if n > kk else -1.0
    assert worst <= tol, (what, "a left-out row beats the k-th", worst)
    assert (s[:, :kk - 1] >= s[:, 1:kk]).all(), (what, "order")
    eq = s[:, :kk - 1] == s[:, 1:kk]
    assert (ids[:, :-1][eq] < ids[:, 1:][eq]).all(), (what, "tie order")
    if k > n:
        assert (i[:, n:] == -1).all() and torch.isinf(s[:, n:]).all(), what


t0 = time.time(); cases = 0; stats = {"fp16": 0, "int8": 0, "many": 0, "flagged": 0}
while time.time() - t0 < secs:
    d = [64, 128, 256, 384, 768, 1024][ri(0, 5)]
    shape = ri(0, 9)
    n = ri(1, 70) if shape == 0 else (ri(64 * 3, 64 * 40) if shape < 4 else ri(20_000, 300_000 if d <= 256 else 120_000))
    if shape == 9:
        n = 256 * ri(4, 300) + [0, 1, 63, 64, 255][ri(0, 4)]
    big = ri(0, 39) == 0                                             # now and then a shard beyond 1 M rows: the select kernel's lists feed the tails
    if big:
        d = [64, 128][ri(0, 1)]; n = ri(1_000_000, 2_600_000)
    nq = [1, ri(2, 64), ri(65, 128), ri(129, 256), ri(257, 1100)][min(4, ri(0, 5))]
    if big:
        nq = min(nq, 300)                                            # (the validator's score matrix is nq x n floats)
    k = [1, 5, 10, 10, 10, 32][ri(0, 5)]
    base = [0, 7, 1 << 33][ri(0, 2)]
    c = torch.nn.functional.normalize(torch.randn((n, d), generator=g, device="cuda"), dim=1)
    q = torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device="cuda"), dim=1)
    if ri(0, 3) == 0:                                                # a common component (anisotropic embeddings; the int8 index is centred on the shard's mean)
        u = torch.nn.functional.normalize(torch.randn((1, d), generator=g, device="cuda"), dim=1) * [0.2, 0.6, 1.5, -0.8][ri(0, 3)]
        c = torch.nn.functional.normalize(c + u, dim=1)
        q = torch.nn.functional.normalize(q + (u if ri(0, 1) else 0.3 * u), dim=1)
    scale = 1.0
    if ri(0, 3) == 0:                                                # rows far from unit norm
        nr = torch.exp(torch.empty(n, device="cuda").uniform_(-4.6, 2.08, generator=g)); c = c * nr[:, None]; scale = float(nr.max().item())
    flavour = ri(0, 5)
    if flavour == 0 and n > 200:                                     # near-ties planted across many groups for query 0
        rows = torch.randperm(n, generator=g, device="cuda")[:min(n, 60)]
        c[rows] = (q[0] * 0.9 + 1e-4 * torch.randn((len(rows), d), generator=g, device="cuda")) * (scale if scale != 1.0 else 1.0)
    elif flavour == 1 and n > 10:                                    # exact duplicates
        rows = torch.randperm(n, generator=g, device="cuda")[:min(n, 300)]; c[rows] = c[rows[0]].clone()
    elif flavour == 2 and n > 64:                                    # a run of neighbours inside one group
        c[64:64 + min(40, n - 64)] = q[min(1, nq - 1)] * 0.8 + 0.02 * torch.randn((min(40, n - 64), d), generator=g, device="cuda")
    c16, q16 = c.to(torch.float16).contiguous(), q.to(torch.float16).contiguous()
    tol = 2e-6 * max(1.0, scale) * 4
    if LOG is not None:                                                # the case about to run, on disk before any kernel of it is launched
        LOG.write(json.dumps(dict(case=cases, n=n, d=d, nq=nq, k=k, base=base, flavour=flavour, scale=round(scale, 3))) + "\n"); LOG.flush()
    for pre in (None, "int8"):
        if pre == "int8" and (d % 128 != 0 or n >= (1 << 32)):
            continue
        idx = ShardIndex(c16, idx_base=base, prefilter=pre, centre_query=[None, None, True, False][ri(0, 3)])      # (int8: the query centred as well, or not, or the index's own choice)
        s, i = idx.search(q16, k)
        what = dict(n=n, d=d, nq=nq, k=k, pre=pre, base=base, flavour=flavour, scale=round(scale, 3), seed=seed, case=cases)
        validate(c16, q16, s, i, k, base, tol, what)
        s2, i2 = idx.search(q16, k)
        assert torch.equal(i, i2) and torch.equal(s, s2), (what, "not repeatable")
        stats["flagged"] += idx.certificate_stats()[0] if pre is None else 0
        if nq <= 1024 and nq >= 4:
            cut = [0, nq // 3, 2 * nq // 3, nq]
            got = idx.search_many([q16[cut[j]:cut[j + 1]] for j in range(3) if cut[j + 1] > cut[j]], k)
            assert torch.equal(torch.cat([x[1] for x in got]), i) and torch.equal(torch.cat([x[0] for x in got]), s), (what, "search_many")
            stats["many"] += 1
        stats["fp16" if pre is None else "int8"] += 1
    cases += 1
    if cases % 50 == 0:
        print(json.dumps({"cases": cases, "seconds": round(time.time() - t0, 1), **stats}), flush=True)
torch.cuda.synchronize()
print(json.dumps({"done": True, "cases": cases, "seconds": round(time.time() - t0, 1), **stats, "seed": seed}))
