"""Encoder weights as a flat {HF state-dict key: float32 ndarray} mapping.

Key names are the ones `transformers` writes for MPNetModel / BertModel
(SURVEY.md §8a-3.3), so a user's local all-mpnet-base-v2 directory loads
without renaming.  `seeded_state_dict` regenerates identical synthetic weights
on any box from (config, seed) — np.random.RandomState streams are bit-stable —
which is how the golden fixtures pin full-size shapes without shipping 109 M
parameters.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict

import numpy as np

from .config import ARCH_MPNET, EncoderConfig


def layer_keys(cfg: EncoderConfig, i: int) -> Dict[str, str]:
    p = f"encoder.layer.{i}."
    if cfg.arch == ARCH_MPNET:
        return {
            "q": p + "attention.attn.q", "k": p + "attention.attn.k", "v": p + "attention.attn.v",
            "o": p + "attention.attn.o", "ln1": p + "attention.LayerNorm",
            "fc1": p + "intermediate.dense", "fc2": p + "output.dense", "ln2": p + "output.LayerNorm",
        }
    return {
        "q": p + "attention.self.query", "k": p + "attention.self.key", "v": p + "attention.self.value",
        "o": p + "attention.output.dense", "ln1": p + "attention.output.LayerNorm",
        "fc1": p + "intermediate.dense", "fc2": p + "output.dense", "ln2": p + "output.LayerNorm",
    }


def expected_shapes(cfg: EncoderConfig) -> Dict[str, tuple]:
    H, F = cfg.hidden, cfg.ffn
    s = {
        "embeddings.word_embeddings.weight": (cfg.vocab_size, H),
        "embeddings.position_embeddings.weight": (cfg.max_pos, H),
        "embeddings.LayerNorm.weight": (H,), "embeddings.LayerNorm.bias": (H,),
    }
    if cfg.arch == ARCH_MPNET:
        s["encoder.relative_attention_bias.weight"] = (cfg.rel_buckets, cfg.heads)
    else:
        s["embeddings.token_type_embeddings.weight"] = (2, H)
    for i in range(cfg.layers):
        k = layer_keys(cfg, i)
        for n in ("q", "k", "v", "o"):
            s[k[n] + ".weight"] = (H, H); s[k[n] + ".bias"] = (H,)
        s[k["fc1"] + ".weight"] = (F, H); s[k["fc1"] + ".bias"] = (F,)
        s[k["fc2"] + ".weight"] = (H, F); s[k["fc2"] + ".bias"] = (H,)
        for n in ("ln1", "ln2"):
            s[k[n] + ".weight"] = (H,); s[k[n] + ".bias"] = (H,)
    return s


def seeded_state_dict(cfg: EncoderConfig, seed: int = 0, std: float = 0.02,
                      bias_std: float = 0.02, ln_jitter: float = 0.0) -> Dict[str, np.ndarray]:
    """Deterministic synthetic weights: N(0, std²) matrices/embeddings, N(0, bias_std²)
    biases, LayerNorm gamma = 1 (+ jitter), beta = 0 (+ jitter).  One RandomState stream,
    keys visited in `expected_shapes` order, so the result depends only on (cfg, seed, stds).

    std=0.02 is the HF init; the fixtures use a larger std so that attention is not
    near-uniform and the relative-position bias / masking actually matter.
    """
    rs = np.random.RandomState(seed)
    sd: Dict[str, np.ndarray] = {}
    for key, shape in expected_shapes(cfg).items():
        if "LayerNorm.weight" in key:
            w = np.ones(shape, np.float32)
            if ln_jitter:
                w = w + (rs.standard_normal(shape) * ln_jitter).astype(np.float32)
        elif "LayerNorm.bias" in key:
            w = np.zeros(shape, np.float32)
            if ln_jitter:
                w = w + (rs.standard_normal(shape) * ln_jitter).astype(np.float32)
        elif key.endswith(".bias"):
            w = (rs.standard_normal(shape) * bias_std).astype(np.float32)
        elif "relative_attention_bias" in key:
            w = (rs.standard_normal(shape) * max(std, 0.5)).astype(np.float32)
        else:
            w = (rs.standard_normal(shape) * std).astype(np.float32)
        sd[key] = w
    return sd


def adversarial_state_dict(cfg: EncoderConfig, seed: int = 0, row_offset: float = 4.0) -> Dict[str, np.ndarray]:
    """Seeded weights with the statistics trained checkpoints have and `seeded_state_dict` lacks — the regime in which a
    LayerNorm folded into the next GEMM as rstd*(acc - mean*s) over a bf16 pre-LN stream could lose precision:
      * LayerNorm gamma log-uniform in [0.25, 4] with two outlier channels at 10 per LayerNorm, beta ~ N(0, 0.5) with the same
        two channels at +-5 (trained BERT/MPNet models carry a few such channels);
      * a common offset of `row_offset` on every output-projection / FFN-2 bias, so each pre-LN row has |mean| well above its
        standard deviation (cancellation in E[y^2] - mean^2, coarse bf16 grid at the row's magnitude);
      * heavy-tailed matrices: Student-t (3 degrees of freedom) entries scaled to std 0.04.
    Same construction on any box from (cfg, seed): fixtures carry only the seed."""
    sd = seeded_state_dict(cfg, seed=seed, std=0.04, bias_std=0.02, ln_jitter=0.0)
    rs = np.random.RandomState(seed + 777)
    for key in list(sd):
        shape = sd[key].shape
        if "LayerNorm.weight" in key:
            g = np.exp(rs.uniform(np.log(0.25), np.log(4.0), size=shape)).astype(np.float32)
            out = rs.choice(shape[0], size=2, replace=False)
            g[out] = 10.0
            sd[key] = g
            b = (rs.standard_normal(shape) * 0.5).astype(np.float32)
            b[out] = np.array([5.0, -5.0], np.float32)
            sd[key.replace(".weight", ".bias")] = b
        elif key.endswith(".weight") and len(shape) == 2 and "embeddings" not in key and "relative_attention_bias" not in key:
            sd[key] = (rs.standard_t(3, size=shape) * (0.04 / np.sqrt(3.0))).astype(np.float32)
    for i in range(cfg.layers):
        k = layer_keys(cfg, i)
        for n in ("o", "fc2"):
            sd[k[n] + ".bias"] = (sd[k[n] + ".bias"] + np.float32(row_offset)).astype(np.float32)
    return sd


def load_hf_dir(path: str | Path, cfg: EncoderConfig) -> Dict[str, np.ndarray]:
    """Load model.safetensors (preferred) or pytorch_model.bin (weights_only=True)
    from a local HF-layout directory; strips an optional 'mpnet.' / 'bert.' prefix."""
    p = Path(path)
    raw: Dict[str, np.ndarray] = {}
    st = p / "model.safetensors"
    if st.exists():
        from safetensors.numpy import load_file
        raw = load_file(str(st))
    elif (p / "pytorch_model.bin").exists():
        import torch
        t = torch.load(str(p / "pytorch_model.bin"), map_location="cpu", weights_only=True)
        raw = {k: v.float().numpy() for k, v in t.items()}
    else:
        raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {p}")
    sd: Dict[str, np.ndarray] = {}
    for k, v in raw.items():
        for pre in ("mpnet.", "bert.", "0.auto_model.", "auto_model."):
            if k.startswith(pre):
                k = k[len(pre):]
        sd[k] = np.ascontiguousarray(v, dtype=np.float32)
    want = expected_shapes(cfg)
    missing = [k for k in want if k not in sd]
    if missing:
        raise KeyError(f"{p}: missing weights {missing[:4]}{'...' if len(missing) > 4 else ''}")
    for k, shp in want.items():
        if tuple(sd[k].shape) != shp:
            raise ValueError(f"{p}: {k} has shape {sd[k].shape}, config expects {shp}")
    return {k: sd[k] for k in want}


def save_hf_dir(path: str | Path, cfg: EncoderConfig, sd: Dict[str, np.ndarray]) -> None:
    """Write a minimal HF-layout directory (config.json + model.safetensors + pooling /
    max_seq_length side files) — used to build synthetic model dirs for tests."""
    import json
    from safetensors.numpy import save_file
    from .config import POOL_CLS
    p = Path(path)
    p.mkdir(parents=True, exist_ok=True)
    hf = {
        "model_type": "mpnet" if cfg.arch == ARCH_MPNET else "bert",
        "vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden,
        "num_hidden_layers": cfg.layers, "num_attention_heads": cfg.heads,
        "intermediate_size": cfg.ffn, "max_position_embeddings": cfg.max_pos,
        "layer_norm_eps": cfg.ln_eps, "pad_token_id": cfg.pad_id,
        "relative_attention_num_buckets": cfg.rel_buckets, "hidden_act": "gelu",
    }
    (p / "config.json").write_text(json.dumps(hf, indent=2))
    (p / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": cfg.max_seq_length}))
    (p / "1_Pooling").mkdir(exist_ok=True)
    (p / "1_Pooling" / "config.json").write_text(json.dumps({
        "word_embedding_dimension": cfg.hidden,
        "pooling_mode_cls_token": cfg.pool == POOL_CLS,
        "pooling_mode_mean_tokens": cfg.pool != POOL_CLS}))
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(p / "model.safetensors"))
