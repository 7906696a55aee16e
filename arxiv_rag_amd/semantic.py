"""Second client of the encoder: the stage-3 semantic chunker.

The reference's `TextProcessor._chunk_semantic` (3-chunks/pipeline/src/processors/text_processor.py:1269-1599)
splits a text into sentences (:1275-1276), encodes them un-normalised with the MiniLM chunking model (:1382-1396),
and walks the sentences once, closing a chunk when the character budget overflows or when the cosine with the
previous sentence drops under 0.7 (:1547-1583); a closed chunk hands its last 20 % of sentences to the next one.

Here the encode and the adjacent-row cosine run on the GPU (`HipSentenceEncoder.encode_device`
then `arx_adjacent_cosine`, so only n-1 floats cross PCIe); the walk itself is character bookkeeping and stays on
the host.  `group_sentences` takes the similarities as an array so it can be checked on CPU against fixtures made by
running the reference's own loop (tools/make_golden.py `semantic`).
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence

SEMANTIC_BREAK = 0.7          # text_processor.py:1560
OVERLAP_FRACTION = 0.2        # text_processor.py:1578
_SENTENCE_END = re.compile(r"(?<=[.!?])\s+")


def split_sentences(text: str) -> List[str]:
    """text_processor.py:1275-1276 — split after ., ! or ? and drop fragments of <= 10 characters (the length test is
    on the un-stripped fragment, as there)."""
    return [s.strip() for s in _SENTENCE_END.split(text) if s.strip() and len(s) > 10]


def group_sentences(sentences: Sequence[str], similarities: Sequence[float], max_chunk_size: int = 1000,
                    min_chunk_size: int = 100, metadata: Optional[Dict] = None,
                    threshold: float = SEMANTIC_BREAK) -> List[Dict]:
    """`similarities[i-1]` = cos(sentence i-1, sentence i).  Same chunks, in the same order, with the same metadata
    keys as text_processor.py:1542-1599."""
    if len(sentences) and len(similarities) != len(sentences) - 1:
        raise ValueError(f"{len(sentences)} sentences need {len(sentences) - 1} similarities, got {len(similarities)}")
    chunks: List[Dict] = []

    def close(members: List[str]):
        text = " ".join(members).strip()
        if len(text) >= min_chunk_size:
            md = dict(metadata or {})
            md.update({"chunk_index": len(chunks), "chunk_method": "semantic"})
            chunks.append({"text": text, "metadata": md})

    members: List[str] = []
    length = 0                                     # characters incl. one joiner per appended sentence (:1583)
    for i, sentence in enumerate(sentences):
        overflow = length + len(sentence) > max_chunk_size
        shift = bool(members) and i > 0 and not overflow and float(similarities[i - 1]) < threshold
        if (overflow or shift) and members:
            close(members)
            keep = min(len(members), max(1, int(len(members) * OVERLAP_FRACTION)))
            members = members[-keep:]
            length = sum(len(s) for s in members)  # the carried sentences are re-counted without joiners (:1580)
        members.append(sentence)
        length += len(sentence) + 1
    if members:
        close(members)
    return chunks


def semantic_chunks(text: str, encoder, max_chunk_size: int = 1000, min_chunk_size: int = 100,
                    metadata: Optional[Dict] = None, batch_size: int = 512) -> Optional[List[Dict]]:
    """Sentence split -> GPU encode -> GPU adjacent cosine -> host grouping.  Returns None where the reference falls
    back to its fixed-size chunker (fewer than two sentences, :1278-1280), which is outside this path."""
    from .encoder import adjacent_cosines
    sentences = split_sentences(text)
    if len(sentences) < 2:
        return None
    emb = encoder.encode_device(sentences, batch_size=batch_size, normalize_embeddings=False)   # :1382-1396, un-normalised
    sims = adjacent_cosines(emb).cpu().numpy()
    return group_sentences(sentences, sims, max_chunk_size, min_chunk_size, metadata)


def semantic_chunks_batch(texts: Sequence[str], encoder, max_chunk_size: int = 1000, min_chunk_size: int = 100,
                          metadatas: Optional[Sequence[Optional[Dict]]] = None, batch_size: int = 512) -> List[Optional[List[Dict]]]:
    """Many documents at once — the MI355X-shaped form of the same step: the sentences of ALL documents go through one
    length-sorted encode (1024-sequence forwards instead of one small forward per paper), one `arx_adjacent_cosine` launch covers
    every adjacent pair (pairs that straddle two documents are computed and ignored), and only the per-document grouping walk runs
    on the host.  Entry i is what `semantic_chunks(texts[i], ...)` returns (None where the reference falls back to fixed chunking)."""
    from .encoder import adjacent_cosines
    split = [split_sentences(t) for t in texts]
    keep = [i for i, s in enumerate(split) if len(s) >= 2]
    out: List[Optional[List[Dict]]] = [None] * len(texts)
    if not keep:
        return out
    flat = [s for i in keep for s in split[i]]
    emb = encoder.encode_device(flat, batch_size=batch_size, normalize_embeddings=False)
    sims = adjacent_cosines(emb).cpu().numpy()
    pos = 0
    for i in keep:
        n = len(split[i])
        md = metadatas[i] if metadatas is not None else None
        out[i] = group_sentences(split[i], sims[pos:pos + n - 1], max_chunk_size, min_chunk_size, md)
        pos += n
    return out
