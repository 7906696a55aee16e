"""Encoder shapes for the embed hot path.

The reference picks a model by Hub name (`SentenceTransformer(model_name)`,
4-embed/generation/generate_embeddings_parallel.py:47; argparse choices :474).
The arithmetic those names stand for is fixed by the checkpoints' config.json;
the shapes below restate them so that synthetic runs and
local-directory loads agree on one description.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, asdict
from pathlib import Path

ARCH_MPNET = 0   # relative-position bias, pad-aware position ids, <s>=0 <pad>=1 </s>=2
ARCH_BERT = 1    # absolute positions + token-type 0, [CLS]/[SEP], [PAD]=0

POOL_MEAN = 0
POOL_CLS = 1


@dataclass(frozen=True)
class EncoderConfig:
    arch: int
    vocab_size: int
    hidden: int
    layers: int
    heads: int
    ffn: int
    max_pos: int
    ln_eps: float
    pool: int
    max_seq_length: int          # sentence-transformers truncation length
    pad_id: int
    rel_buckets: int = 32        # MPNet only
    rel_max_distance: int = 128  # MPNet only

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    def to_json(self) -> str:
        return json.dumps(asdict(self))


# all-mpnet-base-v2: MPNetModel 12L/768, mean pool, max_seq_length 384
MPNET_BASE = EncoderConfig(ARCH_MPNET, 30527, 768, 12, 12, 3072, 514, 1e-5, POOL_MEAN, 384, 1)
# all-MiniLM-L6-v2: BertModel 6L/384 (12 heads of 32), mean pool, max_seq_length 256
MINILM_L6 = EncoderConfig(ARCH_BERT, 30522, 384, 6, 12, 1536, 512, 1e-12, POOL_MEAN, 256, 0)
# BAAI/bge-large-en-v1.5: BertModel 24L/1024, CLS pool, max_seq_length 512
BGE_LARGE = EncoderConfig(ARCH_BERT, 30522, 1024, 24, 16, 4096, 512, 1e-12, POOL_CLS, 512, 0)

# tiny shapes used by the golden fixtures (tests/golden) — small enough to commit whole
TINY_MPNET = EncoderConfig(ARCH_MPNET, 128, 64, 2, 2, 128, 66, 1e-5, POOL_MEAN, 64, 1)
TINY_BERT = EncoderConfig(ARCH_BERT, 128, 64, 2, 2, 128, 64, 1e-12, POOL_MEAN, 64, 0)
TINY_BERT_CLS = EncoderConfig(ARCH_BERT, 128, 128, 2, 2, 256, 64, 1e-12, POOL_CLS, 64, 0)

PRESETS = {
    "all-mpnet-base-v2": MPNET_BASE,
    "sentence-transformers/all-mpnet-base-v2": MPNET_BASE,
    "all-MiniLM-L6-v2": MINILM_L6,
    "sentence-transformers/all-MiniLM-L6-v2": MINILM_L6,
    "BAAI/bge-large-en-v1.5": BGE_LARGE,
    "bge-large-en-v1.5": BGE_LARGE,
    "tiny-mpnet": TINY_MPNET,
    "tiny-bert": TINY_BERT,
    "tiny-bert-cls": TINY_BERT_CLS,
}


def config_from_hf_dir(path: str | Path) -> EncoderConfig:
    """Read an HF / sentence-transformers model directory (config.json,
    sentence_bert_config.json, 1_Pooling/config.json)."""
    p = Path(path)
    cfg = json.loads((p / "config.json").read_text())
    mt = cfg.get("model_type", "")
    if mt == "mpnet":
        arch, pad = ARCH_MPNET, cfg.get("pad_token_id", 1)
    elif mt == "bert":
        arch, pad = ARCH_BERT, cfg.get("pad_token_id", 0)
    else:
        raise ValueError(f"unsupported model_type {mt!r} in {p/'config.json'}")
    pool = POOL_MEAN
    pj = p / "1_Pooling" / "config.json"
    if pj.exists():
        pc = json.loads(pj.read_text())
        if pc.get("pooling_mode_cls_token"):
            pool = POOL_CLS
    msl = min(cfg["max_position_embeddings"] - (2 if arch == ARCH_MPNET else 0), 512)
    sj = p / "sentence_bert_config.json"
    if sj.exists():
        msl = json.loads(sj.read_text()).get("max_seq_length", msl)
    return EncoderConfig(
        arch=arch, vocab_size=cfg["vocab_size"], hidden=cfg["hidden_size"],
        layers=cfg["num_hidden_layers"], heads=cfg["num_attention_heads"],
        ffn=cfg["intermediate_size"], max_pos=cfg["max_position_embeddings"],
        ln_eps=cfg.get("layer_norm_eps", 1e-12), pool=pool, max_seq_length=msl, pad_id=pad,
        rel_buckets=cfg.get("relative_attention_num_buckets", 32),
    )
