"""Hand-off from the stage-4 output files to the HBM-resident index.

* `load_embeddings_from_disk` reads what stage 4 leaves on disk — either the layout GEN writes
  (`embeddings.npy` + `metadata.json` + `index.json`, generate_embeddings_parallel.py:271-321) or the batched
  layout of `4-embed/utils/save_embeddings_to_disk.py:15-80` (`embeddings_batch_%04d.npy`,
  `metadata_batch_%04d.json`, `index.json` with `num_batches`), with the same return value as that file's
  `load_embeddings_from_disk` (:82-117): `(embeddings [N, D], metadata list)`.
* `save_embeddings_disk` writes that batched layout (byte-identical to the reference's writer).
* `HipCollection` keeps this rank's rows in HBM as fp16 and answers `query(...)` in the shape of a Chroma
  collection (`ids`, `documents`, `metadatas`, `distances` per query), so code written against the collection the
  reference fills at GEN:404-424 ports over.  Distances are squared L2 (Chroma's default space), which on unit
  rows is `2 - 2*cosine`: the same ranking as the cosine top-k the kernels compute.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def load_embeddings_from_disk(input_dir: str | Path, batch_index: Optional[int] = None,
                              mmap: bool = True) -> Tuple[np.ndarray, List[Dict]]:
    p = Path(input_dir)
    if batch_index is not None:
        emb = np.load(p / f"embeddings_batch_{batch_index:04d}.npy", mmap_mode="r" if mmap else None)
        meta = json.loads((p / f"metadata_batch_{batch_index:04d}.json").read_text(encoding="utf-8"))
        return emb, meta
    index = json.loads((p / "index.json").read_text(encoding="utf-8"))
    if "num_batches" in index:                               # batched layout
        embs, meta = [], []
        for i in range(index["num_batches"]):
            embs.append(np.load(p / f"embeddings_batch_{i:04d}.npy", mmap_mode="r" if mmap else None))
            meta.extend(json.loads((p / f"metadata_batch_{i:04d}.json").read_text(encoding="utf-8")))
        return np.vstack(embs), meta
    emb = np.load(p / "embeddings.npy", mmap_mode="r" if mmap else None)      # GEN layout
    meta = json.loads((p / "metadata.json").read_text(encoding="utf-8"))
    if emb.shape[0] != index.get("total_embeddings", emb.shape[0]):
        raise ValueError(f"{p}: index.json says {index['total_embeddings']} rows, embeddings.npy has {emb.shape[0]}")
    return emb, meta


def save_embeddings_disk(chunks: Sequence[Dict], embeddings, output_dir: str | Path = "./embeddings_saved", batch_size: int = 10000) -> None:
    """Writer of the BATCHED on-disk layout (`4-embed/utils/save_embeddings_to_disk.py:15-80`, the reference's alternative to the three
    files GEN writes): `embeddings_batch_%04d.npy` (float64 — the reference's `.tolist()` round trip — [<= batch_size, D]),
    `metadata_batch_%04d.json` (the GEN fields + `batch_index`, `batch_position`; indent 2, ensure_ascii=False) and `index.json`
    (`total_embeddings`, `embedding_dimension`, `num_batches`, `batch_size`, `chunks` = every chunk's id or null).  Byte-identical to
    what the reference's function writes (tests/golden/harness/expected_batched, produced by that function).  `embeddings`: anything
    indexable by row ([N, D] array, memmap, list of rows); one batch is in memory at a time."""
    out = Path(output_dir)
    out.mkdir(parents=True, exist_ok=True)
    n = len(embeddings)
    num_batches = (n + batch_size - 1) // batch_size
    dim = int(np.asarray(embeddings[0]).shape[0]) if n else 0
    for i in range(num_batches):
        a, b = i * batch_size, min((i + 1) * batch_size, n)
        np.save(out / f"embeddings_batch_{i:04d}.npy", np.asarray(embeddings[a:b], dtype=np.float64))
        meta = []
        for j, ch in enumerate(chunks[a:b]):
            m = ch.get("metadata", {})
            meta.append({"chunk_id": ch.get("chunk_id", f"chunk_{a + j}"), "paper_id": m.get("paper_id"), "section": m.get("section"),
                         "quality_score": m.get("quality_score"), "text": ch["text"], "text_length": len(ch["text"]),
                         "batch_index": i, "batch_position": j})
        with open(out / f"metadata_batch_{i:04d}.json", "w", encoding="utf-8") as fh:
            json.dump(meta, fh, indent=2, ensure_ascii=False)
    index = {"total_embeddings": n, "embedding_dimension": dim, "num_batches": num_batches, "batch_size": batch_size,
             "chunks": [ch.get("chunk_id") for ch in chunks]}
    with open(out / "index.json", "w", encoding="utf-8") as fh:
        json.dump(index, fh, indent=2)


class HipCollection:
    """This rank's shard of the corpus, resident in HBM, with a Chroma-shaped `query`."""

    def __init__(self, embeddings: np.ndarray, metadata: Sequence[Dict], device="cuda:0", encoder=None,
                 rank: int = 0, world: int = 1, chunk_rows: int = 1 << 18):
        import torch
        from .index import ShardIndex, shard_bounds
        n, d = embeddings.shape
        lo, hi = shard_bounds(n, world, rank)
        self.n_total, self.dim, self.lo, self.hi = n, d, lo, hi
        self.metadata = metadata
        self.encoder = encoder
        shard = torch.empty((hi - lo, d), dtype=torch.float16, device=device)
        for s0 in range(lo, hi, chunk_rows):                   # stream: never a second full copy in host RAM
            s1 = min(hi, s0 + chunk_rows)
            shard[s0 - lo:s1 - lo] = torch.from_numpy(np.ascontiguousarray(embeddings[s0:s1], dtype=np.float16)).to(device)
        dim = int(shard.shape[1]) if shard.dim() == 2 else 0
        self.index = ShardIndex(shard, idx_base=lo, prefilter="int8" if (dim % 128 == 0 and 0 < dim <= 1024 and shard.shape[0] > 0) else None,
                                adaptive=True)

    @classmethod
    def from_disk(cls, input_dir, **kw) -> "HipCollection":
        emb, meta = load_embeddings_from_disk(input_dir)
        return cls(emb, meta, **kw)

    def count(self) -> int:
        return self.n_total

    def query(self, query_embeddings=None, query_texts: Optional[Sequence[str]] = None, n_results: int = 10) -> Dict:
        import torch
        if query_embeddings is None:
            if query_texts is None or self.encoder is None:
                raise ValueError("pass query_embeddings, or query_texts with an encoder")
            query_embeddings = self.encoder.encode(list(query_texts), normalize_embeddings=True, convert_to_numpy=True, low_latency=True)
        q = torch.from_numpy(np.ascontiguousarray(query_embeddings, dtype=np.float16)).to(self.index.corpus.device)
        if q.dim() == 1:
            q = q[None]
        s, i = self.index.search_distributed(q, n_results)
        s, i = s.cpu().numpy(), i.cpu().numpy()
        out = {"ids": [], "distances": [], "scores": [], "documents": [], "metadatas": [], "indices": []}
        for qi in range(q.shape[0]):
            keep = i[qi] >= 0
            rows = i[qi][keep].tolist()
            ms = [self.metadata[r] for r in rows]
            out["indices"].append(rows)
            out["ids"].append([m.get("chunk_id", f"chunk_{r}") for m, r in zip(ms, rows)])
            out["scores"].append(s[qi][keep].tolist())
            out["distances"].append((2.0 - 2.0 * s[qi][keep]).tolist())
            out["documents"].append([m.get("text") for m in ms])
            out["metadatas"].append([{k: m.get(k) for k in ("paper_id", "section", "quality_score")} for m in ms])
        return out
