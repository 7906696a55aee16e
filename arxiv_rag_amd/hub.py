"""Resolve a model NAME to a LOCAL directory and build the encoder from it.

The reference loads by Hub name (`SentenceTransformer(model_name)`, generate_embeddings_parallel.py:47),
which downloads on first use.  Here nothing is ever fetched: a name resolves to
  1. the path itself if it is a directory,
  2. $ARX_MODEL_DIR/<name> (also with '/' replaced by '--'),
  3. an already-populated HF cache snapshot (~/.cache/huggingface/hub/models--<org>--<name>/snapshots/*),
and otherwise raises FileNotFoundError (the CLI turns that into exit code 1, like a missing dependency).
"""
from __future__ import annotations

import os
from pathlib import Path

from .config import PRESETS, config_from_hf_dir

_ORGS = {"all-mpnet-base-v2": "sentence-transformers", "all-MiniLM-L6-v2": "sentence-transformers",
         "bge-large-en-v1.5": "BAAI"}


def resolve_model_dir(name: str) -> Path:
    p = Path(name)
    if p.is_dir() and (p / "config.json").exists():
        return p
    cands = []
    root = os.environ.get("ARX_MODEL_DIR")
    if root:
        cands += [Path(root) / name, Path(root) / name.replace("/", "--"), Path(root) / name.split("/")[-1]]
    short = name.split("/")[-1]
    org = name.split("/")[0] if "/" in name else _ORGS.get(short, "sentence-transformers")
    hub = Path(os.environ.get("HF_HOME", Path.home() / ".cache" / "huggingface")) / "hub" / f"models--{org}--{short}" / "snapshots"
    if hub.is_dir():
        cands += sorted(hub.iterdir())
    for c in cands:
        if c.is_dir() and (c / "config.json").exists():
            return c
    raise FileNotFoundError(
        f"model {name!r}: no local directory found (looked at {[str(c) for c in cands] or 'nothing'}). "
        f"Pass a directory, or set ARX_MODEL_DIR; models are never downloaded.")


def load_sentence_encoder(name: str, device="cuda:0", max_batch: int = 1024):
    """Name or directory -> HipSentenceEncoder (tokenizer + HIP encoder)."""
    from .encoder import HipSentenceEncoder
    from .tokenizer import WordPieceTokenizer
    from .weights import load_hf_dir
    d = resolve_model_dir(name)
    cfg = config_from_hf_dir(d)
    preset = PRESETS.get(name) or PRESETS.get(name.split("/")[-1])
    if preset is not None and (preset.hidden, preset.layers, preset.arch) != (cfg.hidden, cfg.layers, cfg.arch):
        raise ValueError(f"{d} does not hold a {name} checkpoint (shape mismatch)")
    tdir = d / "0_Transformer" if (d / "0_Transformer" / "config.json").exists() else d
    sd = load_hf_dir(tdir, cfg)
    tok = WordPieceTokenizer.from_dir(tdir, cfg)
    return HipSentenceEncoder(cfg, sd, tok, device=device, max_batch=max_batch)
