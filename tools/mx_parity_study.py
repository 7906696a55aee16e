#!/usr/bin/env python3
"""configs[4] groundwork (BASELINE.json: bge-large, "fp8 MFMA encode"): how much of the north-star parity budget
(per-vector cosine >= 1 - 1e-3 against the fp32 reference) does rounding the GEMM INPUTS to 8 bits spend?

Measured before anything is built.  On the committed golden fixtures (outputs of transformers' fp32 modules:
tests/golden/full_shapes.npz and full_shapes_adv.npz) a compact fp32 restatement of the encoder is run with the inputs of a chosen
subset of the four linear layers {qkv, o, fc1, fc2} rounded to
    bf16      what the shipped HIP path does (baseline row; its GEMM outputs / residual stream are rounded to bf16 in every mode)
    e4m3      plain OCP e4m3fn, one power-of-two scale per TENSOR (the non-scaled fp8 MFMA: bf16 rate on gfx950 — no speed-up, listed for reference)
    mxfp8     OCP MX: blocks of 32 elements along K share an e8m0 (power-of-two) scale, elements e4m3fn — the format
              v_mfma_scale_f32_*_f8f6f4 consumes at 2x the bf16 rate (MI355X_MICROARCH.md, Matrix cores)
and the worst 1 - cos over the fixture's sequences is reported.  Accumulation is fp32 in every mode (as on the MFMA).
Attention's QK^T / PV products, softmax, LayerNorm and the residual adds are left as the bf16 path has them.

  python tools/mx_parity_study.py [--quick]        ->  markdown table on stdout (pasted into DESIGN.md), JSON to profiles/r02/

CPU only (torch); no oracle import: the restatement below is the study's own.
"""
from __future__ import annotations

import itertools
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from arxiv_rag_amd import config as C                                       # noqa: E402
from arxiv_rag_amd.weights import adversarial_state_dict, layer_keys, seeded_state_dict   # noqa: E402

LINEARS = ("qkv", "o", "fc1", "fc2")


def q_bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def q_e4m3_tensor(x):
    """plain e4m3fn with one power-of-two scale per tensor (amax -> [256, 448])"""
    amax = x.abs().max().clamp_min(1e-30)
    s = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
    return (x / s).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32) * s


def q_mxfp8(x):
    """OCP MX fp8: per 32-element block along the LAST (= K) axis, X = 2^(floor(log2(amax)) - 8), elements e4m3fn (saturating)"""
    shp = x.shape
    K = shp[-1]
    assert K % 32 == 0
    b = x.reshape(-1, K // 32, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -120)
    s = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
    q = (b / s).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32) * s
    return q.reshape(shp)


QUANT = {"bf16": q_bf16, "e4m3": q_e4m3_tensor, "mxfp8": q_mxfp8}


class Enc:
    """fp32 restatement (torch CPU) of the MPNet / BERT encoder + pool + L2 with a hook on every linear layer's inputs."""

    def __init__(self, cfg, sd):
        self.cfg = cfg
        self.sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
        self.wq = {}                                 # (mode, layer, name) -> quantised weight, cached

    def lin(self, x, li, name, mode_of):
        k = layer_keys(self.cfg, li)
        if name == "qkv":
            W = torch.cat([self.sd[k[n] + ".weight"] for n in ("q", "k", "v")], 0)
            b = torch.cat([self.sd[k[n] + ".bias"] for n in ("q", "k", "v")], 0)
        else:
            W, b = self.sd[k[name] + ".weight"], self.sd[k[name] + ".bias"]
        mode = mode_of(name)
        key = (mode, li, name)
        if key not in self.wq:
            self.wq[key] = QUANT[mode](W)
        return q_bf16(QUANT[mode](x) @ self.wq[key].T + b)          # stored activation = bf16, as in the HIP path

    @staticmethod
    def ln(x, g, b, eps):
        mu = x.mean(-1, keepdim=True)
        var = ((x - mu) ** 2).mean(-1, keepdim=True)
        return (x - mu) / torch.sqrt(var + eps) * g + b

    def forward(self, ids, lens, mode_of):
        cfg, sd = self.cfg, self.sd
        ids_t = torch.from_numpy(ids.astype(np.int64))
        B, S = ids_t.shape
        lens_t = torch.from_numpy(lens.astype(np.int64))
        mask = torch.arange(S)[None, :] < lens_t[:, None]
        if cfg.arch == C.ARCH_MPNET:
            npad = (ids_t != cfg.pad_id).long()
            pos = torch.cumsum(npad, 1) * npad + cfg.pad_id
            x = sd["embeddings.word_embeddings.weight"][ids_t] + sd["embeddings.position_embeddings.weight"][pos]
        else:
            x = sd["embeddings.word_embeddings.weight"][ids_t] + sd["embeddings.position_embeddings.weight"][:S][None] \
                + sd["embeddings.token_type_embeddings.weight"][0]
        x = q_bf16(self.ln(x, sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], cfg.ln_eps))
        H, nh = cfg.hidden, cfg.heads
        dh = H // nh
        bias = None
        if cfg.arch == C.ARCH_MPNET:
            rel = torch.arange(S)[None, :] - torch.arange(S)[:, None]
            bucket = mpnet_buckets(rel, cfg.rel_buckets, cfg.rel_max_distance)
            bias = sd["encoder.relative_attention_bias.weight"][bucket].permute(2, 0, 1)[None]      # [1, nh, S, S]
        neg = torch.zeros((B, 1, 1, S)).masked_fill(~mask[:, None, None, :], float("-inf"))
        for li in range(cfg.layers):
            k = layer_keys(cfg, li)
            qkv = self.lin(x, li, "qkv", mode_of)
            q, kk, v = [t.reshape(B, S, nh, dh).transpose(1, 2) for t in qkv.split(H, -1)]
            sc = q @ kk.transpose(-1, -2) / np.sqrt(dh) + neg
            if bias is not None:
                sc = sc + bias
            p = q_bf16(torch.softmax(sc, -1))
            ctx = q_bf16((p @ v).transpose(1, 2).reshape(B, S, H))
            y1 = q_bf16(self.lin(ctx, li, "o", mode_of) + x)
            x1 = self.ln(y1, sd[k["ln1"] + ".weight"], sd[k["ln1"] + ".bias"], cfg.ln_eps)
            hmid = self.lin(q_bf16(x1), li, "fc1", mode_of)
            hmid = q_bf16(torch.nn.functional.gelu(hmid))
            y2 = q_bf16(self.lin(hmid, li, "fc2", mode_of) + x1)
            x = q_bf16(self.ln(y2, sd[k["ln2"] + ".weight"], sd[k["ln2"] + ".bias"], cfg.ln_eps))
        if cfg.pool == C.POOL_CLS:
            pooled = x[:, 0]
        else:
            mf = mask[..., None].float()
            pooled = (x * mf).sum(1) / mf.sum(1).clamp_min(1e-9)
        return torch.nn.functional.normalize(pooled, dim=1).numpy()


def mpnet_buckets(rel, num_buckets=32, max_distance=128):
    """transformers modeling_mpnet.py relative_position_bucket (bidirectional), rel = key - query"""
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float().clamp_min(1) / max_exact) / np.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def fixtures(quick):
    g = np.load(ROOT / "tests" / "golden" / "full_shapes.npz")
    ga = np.load(ROOT / "tests" / "golden" / "full_shapes_adv.npz")
    out = []
    for name, key, src in (("all-mpnet-base-v2", "all-mpnet-base-v2:w05", g), ("all-mpnet-base-v2", "all-mpnet-base-v2:hf02", g),
                           ("all-mpnet-base-v2", "all-mpnet-base-v2:adv", ga),
                           ("BAAI/bge-large-en-v1.5", "BAAI_bge-large-en-v1.5:w05", g), ("BAAI/bge-large-en-v1.5", "BAAI_bge-large-en-v1.5:adv", ga)):
        cfg = C.PRESETS[name]
        if key.endswith(":adv"):
            seed, off = src[key + ":wspec"]
            sd = adversarial_state_dict(cfg, seed=int(seed), row_offset=float(off))
        else:
            seed, std, bstd, jit = src[key + ":wspec"]
            sd = seeded_state_dict(cfg, seed=int(seed), std=std, bias_std=bstd, ln_jitter=jit)
        ids, lens, ref = src[key + ":ids"], src[key + ":lens"], src[key + ":emb"]
        if quick:
            ids, lens, ref = ids[:6], lens[:6], ref[:6]
        out.append((key, cfg, sd, ids, lens, ref))
    return out


def main():
    quick = "--quick" in sys.argv
    torch.set_grad_enabled(False)
    subsets = [(), ("qkv",), ("o",), ("fc1",), ("fc2",), ("qkv", "o"), ("fc1", "fc2"), LINEARS]
    results = {}
    t0 = time.time()
    for key, cfg, sd, ids, lens, ref in fixtures(quick):
        enc = Enc(cfg, sd)
        for mode, sub in itertools.product(("e4m3", "mxfp8"), subsets):
            if not sub and mode != "e4m3":
                continue                              # the empty subset is the bf16 baseline: once
            emb = enc.forward(ids, lens, lambda n, sub=sub, mode=mode: mode if n in sub else "bf16")
            cos = (emb * ref).sum(1) / (np.linalg.norm(emb, axis=1) * np.linalg.norm(ref, axis=1))
            tag = "bf16 (baseline)" if not sub else f"{mode}: " + "+".join(sub)
            results.setdefault(key, {})[tag] = float(1 - cos.min())
            print(f"[{time.time() - t0:6.0f}s] {key:34s} {tag:28s} worst 1-cos = {1 - cos.min():.2e}", file=sys.stderr)
        del enc
    tags = list(next(iter(results.values())).keys())
    print("| GEMM inputs rounded to | " + " | ".join(results.keys()) + " |")
    print("|---|" + "---|" * len(results))
    for t in tags:
        print(f"| {t} | " + " | ".join(f"{results[k][t]:.1e}" for k in results) + " |")
    out = ROOT / "profiles" / "r02"
    out.mkdir(parents=True, exist_ok=True)
    (out / ("mx_parity_study_quick.json" if quick else "mx_parity_study.json")).write_text(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
