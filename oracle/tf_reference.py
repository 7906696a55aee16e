"""ORACLE — test infrastructure, not product code.

The third-party arithmetic itself: `transformers.MPNetModel` / `BertModel` (what sentence-transformers chains under
`SentenceTransformer.encode`, generate_embeddings_parallel.py:146-153) built from an explicit LOCAL config with the
given weights, plus the mean/CLS pool and L2 normalise tail, fp32 eager on CPU.  Used by tools/make_golden.py to
generate the golden vectors and by bench.py's cpu_baseline leg as the "reference-equivalent CPU" timing
(CPU-A).  Only tests/, tools/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import numpy as np


def build_model(cfg, sd):
    import torch
    import transformers
    if cfg.arch == 0:
        hc = transformers.MPNetConfig(
            vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
            num_attention_heads=cfg.heads, intermediate_size=cfg.ffn, max_position_embeddings=cfg.max_pos,
            layer_norm_eps=cfg.ln_eps, relative_attention_num_buckets=cfg.rel_buckets,
            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        m = transformers.MPNetModel(hc, add_pooling_layer=False)
    else:
        hc = transformers.BertConfig(
            vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
            num_attention_heads=cfg.heads, intermediate_size=cfg.ffn, max_position_embeddings=cfg.max_pos,
            layer_norm_eps=cfg.ln_eps, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
            attn_implementation="eager")
        m = transformers.BertModel(hc, add_pooling_layer=False)
    res = m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    return m.eval()


def encode_tokens(model, cfg, ids: np.ndarray, lens: np.ndarray) -> np.ndarray:
    import torch
    ids_t = torch.from_numpy(np.asarray(ids, np.int64))
    S = ids_t.shape[1]
    mask = (torch.arange(S)[None, :] < torch.from_numpy(np.asarray(lens, np.int64))[:, None]).long()
    with torch.no_grad():
        h = model(input_ids=ids_t, attention_mask=mask, return_dict=True).last_hidden_state
        if cfg.pool == 1:
            pooled = h[:, 0]
        else:
            mf = mask.unsqueeze(-1).float()
            pooled = (h * mf).sum(1) / torch.clamp(mf.sum(1), min=1e-9)
        return torch.nn.functional.normalize(pooled, p=2, dim=1).numpy()
