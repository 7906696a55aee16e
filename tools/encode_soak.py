#!/usr/bin/env python3
"""Race / edge screen of the encode path (the counterpart of tools/search_soak.py and tools/gemm_soak.py).  The encoder promises that a row
does not depend on what else is in its forward (token-packed activations, fixed-order row statistics, one attention block per (sequence,
head): DESIGN.md §3) — BITWISE.  That is a size-independent property the CPU oracle is not needed for: a pool of random token sequences
(lengths 1 ... max_seq_length, weighted towards the tile edges 1/2/31/32/33/63/64/65/127/128/129/255/256/257/383/384/511/512) is encoded
once in a canonical batching, then for `seconds` per model random subsets in random order and random batch sizes (1 ... 1 024 sequences: token
counts ragged against every GEMM tile, forwards from one token to ~260 k) must reproduce the canonical rows bit for bit, f32 and fp16
outputs both; rows are also checked finite and unit-norm.  A wave that reads a neighbour's tile, a tail tile stored past its rows, a
missing wait in the LDS-DMA ring or a stale workspace between forwards of different sizes all show up as a changed bit.
usage: encode_soak.py [seconds per model] [seed]"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from arxiv_rag_amd import config as C
from arxiv_rag_amd.encoder import HipEncoder
from arxiv_rag_amd.weights import seeded_state_dict

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
EDGES = [1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 383, 384, 385, 511, 512]
report = {}
for name in ["all-mpnet-base-v2", "all-MiniLM-L6-v2", "BAAI/bge-large-en-v1.5"]:
    cfg = C.PRESETS[name]
    rs = np.random.RandomState(seed)
    sd = seeded_state_dict(cfg, seed=seed + 10, std=0.04, bias_std=0.02, ln_jitter=0.05)
    enc = HipEncoder(cfg, sd)
    L = cfg.max_seq_length
    P = 2048
    lens = np.where(rs.rand(P) < 0.35, rs.choice([e for e in EDGES if e <= L], size=P), rs.randint(1, L + 1, size=P))
    lens[:8] = [L, 1, L, 2, L - 1, 1, 1, L]
    pool = [rs.randint(4, cfg.vocab_size - 1, size=int(n)).tolist() for n in lens]
    ref16 = torch.zeros((P, cfg.hidden), dtype=torch.float16, device="cuda")
    ref = enc.encode_ragged(pool, batch_size=256, on_device=True, out_f16=ref16).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all(), name
    nrm = ref.norm(dim=1)
    assert (nrm - 1).abs().max().item() < 1e-5, (name, "unit norm", (nrm - 1).abs().max().item())
    # the canonical pass itself, repeated: bit-repeatable run to run
    again = enc.encode_ragged(pool, batch_size=256, on_device=True)
    assert torch.equal(again, ref), (name, "not repeatable")
    t0 = time.time(); trials = 0; tokens = 0; smallest = 1 << 30; largest = 0
    while time.time() - t0 < secs:
        mode = rs.randint(0, 6)
        n = [1, rs.randint(1, 9), rs.randint(2, 65), rs.randint(65, 300), rs.randint(300, 1200), rs.randint(1, P + 1)][mode]
        pick = rs.choice(P, size=n, replace=(rs.rand() < 0.2))       # with repeats now and then: the same row twice in one forward
        bs = [1, rs.randint(1, 33), rs.randint(33, 257), rs.randint(257, 1025), 1024][rs.randint(0, 5)]
        if bs == 1 and n > 64:
            bs = 7
        seqs = [pool[i] for i in pick]
        o16 = torch.full((n, cfg.hidden + (8 if rs.rand() < 0.3 else 0)), 7.0, dtype=torch.float16, device="cuda")     # a wider row pitch now and then
        out = enc.encode_ragged(seqs, batch_size=int(bs), on_device=True, out_f16=o16)
        pk = torch.from_numpy(pick).cuda()
        if not torch.equal(out, ref[pk]):
            bad = (out != ref[pk]).any(dim=1).nonzero().flatten()[:8].tolist()
            raise SystemExit(f"{name}: f32 rows differ from the canonical batching: n={n} bs={bs} rows {bad} lens {[len(seqs[b]) for b in bad]}")
        if not torch.equal(o16[:, :cfg.hidden], ref16[pk]):
            raise SystemExit(f"{name}: fp16 rows differ from the canonical batching: n={n} bs={bs}")
        if o16.shape[1] > cfg.hidden:
            assert (o16[:, cfg.hidden:] == 7.0).all(), (name, "wrote beyond the row's hidden columns")
        tk = int(sum(len(s) for s in seqs))
        trials += 1; tokens += tk; smallest = min(smallest, tk); largest = max(largest, tk)
    torch.cuda.synchronize()
    enc.close()
    report[name] = {"trials": trials, "tokens": tokens, "smallest_call_tokens": smallest, "largest_call_tokens": largest, "seconds": round(time.time() - t0, 1)}
    print(name, report[name], flush=True)
print(json.dumps({"encode_soak": "green", "seed": seed, "models": report}))
