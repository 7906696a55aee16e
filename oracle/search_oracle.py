"""ORACLE — test infrastructure, not product code.

Brute-force cosine top-k.  The reference has NO search code:
the only pins are `top_k: 10` (/root/reference/3-chunks/pipeline/config.yaml:64) and the
two-vector cosine helper (/root/reference/3-chunks/pipeline/src/processors/
text_processor.py:1601-1605).  The definition used here is the one SURVEY.md §8c fixes:

    S   = Q32 @ C32.T         fp32 numpy on the fp16-ROUNDED inputs (unit rows => dot = cosine)
    idx = argsort(-S, axis=1, kind='stable')[:, :k]      ties -> lower row index

"parity unpinned": no reference test or fixture covers this step.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np


def topk_search(corpus_f16: np.ndarray, queries_f16: np.ndarray, k: int = 10, idx_base: int = 0,
                block: int = 65536):
    """corpus [N,D] float16, queries [Q,D] float16 -> (scores f32 [Q,k], ids int64 [Q,k]).
    Rows beyond N (k > N) are (-inf, -1).  Blocked over N so 10 M rows never materialise Q×N."""
    C = np.asarray(corpus_f16)
    Q = np.asarray(queries_f16).astype(np.float32)
    N, nq = C.shape[0], Q.shape[0]
    best_s = np.full((nq, k), -np.inf, np.float32)
    best_i = np.full((nq, k), -1, np.int64)
    for b0 in range(0, N, block):
        Cb = C[b0:b0 + block].astype(np.float32)
        S = Q @ Cb.T
        ids = np.arange(b0, b0 + Cb.shape[0], dtype=np.int64)[None, :].repeat(nq, 0) + idx_base
        best_s, best_i = merge_topk(np.concatenate([best_s, S], 1), np.concatenate([best_i, ids], 1), k)
    return best_s, best_i


def merge_topk(scores: np.ndarray, ids: np.ndarray, k: int):
    """Per row: the k best (score desc, id asc on ties; id -1 = empty, always last)."""
    ids_key = np.where(ids < 0, np.iinfo(np.int64).max, ids)
    order = np.lexsort((ids_key, -scores.astype(np.float64)), axis=1)[:, :k]
    return np.take_along_axis(scores, order, 1).astype(np.float32), np.take_along_axis(ids, order, 1)


def merge_partials(scores: np.ndarray, ids: np.ndarray, k: int):
    """[P,Q,k] per-shard partial top-k (global ids) -> [Q,k]; the step after the all-gather."""
    P, nq, kk = scores.shape
    return merge_topk(scores.transpose(1, 0, 2).reshape(nq, P * kk),
                      ids.transpose(1, 0, 2).reshape(nq, P * kk), k)


def unit_rows_f16(n: int, d: int, seed: int) -> np.ndarray:
    """L2-normalised N(0,1) rows rounded to fp16 (bench cfg 3 recipe)."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float16)
