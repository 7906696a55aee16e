#!/usr/bin/env python3
"""Drop-in for stage 4 of arxiv-rag: `4-embed/generation/generate_embeddings_parallel.py` ("GEN").

Same positional argument, flags and defaults (GEN:471-491), same exit codes (GEN:497-499, 527-529, 603-609),
same `./embeddings_saved/{embeddings.npy, metadata.json, index.json}` layout (GEN:271-321), same
never-raise-per-chunk policy (GEN:155-177).  What changes is underneath: one process per MI355X
(`torchrun --nproc-per-node N`), each rank encodes a contiguous shard of the chunk list through the HIP
encoder and keeps its rows in HBM as its fp16 corpus shard; an optional `--queries` step then serves
brute-force cosine top-k over the shards (RCCL all-gather of partial top-k).

Additive flags only: --model also takes a local directory; --model-dir, --out-dtype, --queries, --top-k,
--skip-chroma.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import threading
import time
import traceback
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

MODEL_CHOICES = ["all-mpnet-base-v2", "all-MiniLM-L6-v2"]          # GEN:474

_model = None
_model_name = None


# --------------------------------------------------------------------------------------------- model
def init_worker_model(model_name: str, factory: Optional[Callable] = None):
    """Per-process singleton (GEN:40-65): here one process owns one GPU."""
    global _model, _model_name
    if _model is None or _model_name != model_name:
        if factory is None:
            from .hub import load_sentence_encoder
            dev = f"cuda:{os.environ.get('LOCAL_RANK', '0')}"
            factory = lambda name: load_sentence_encoder(name, device=dev)     # noqa: E731
        print(f"[rank {os.environ.get('RANK', '0')}] Loading model: {model_name}")
        _model = factory(model_name)
        _model_name = model_name
        print(f"[rank {os.environ.get('RANK', '0')}] Model loaded successfully")
    return _model


def get_worker_model(model_name: str):
    """GEN:67-74."""
    return init_worker_model(model_name)


# --------------------------------------------------------------------------------------------- load
def load_chunks_from_file(file_path: Path, min_quality: float = 0.8) -> List[Dict]:
    """One chunk-JSON file -> its chunks with metadata.quality_score >= min_quality (missing = 0);
    unreadable files contribute nothing (GEN:76-92)."""
    kept: List[Dict] = []
    try:
        with open(file_path, "r", encoding="utf-8") as fh:
            doc = json.load(fh)
        for ch in doc.get("chunks", []):
            if ch.get("metadata", {}).get("quality_score", 0) >= min_quality:
                kept.append(ch)
    except Exception:
        pass
    return kept


def _load_many(args: Tuple[List[str], float]) -> List[Dict]:
    paths, min_quality = args
    out: List[Dict] = []
    for p in paths:
        out.extend(load_chunks_from_file(Path(p), min_quality))
    return out


PROCESS_POOL_MIN_FILES = 4096      # below this a spawn pool's start-up (~0.2 s per worker) costs more than it saves


def effective_cpus() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota.  `mp.cpu_count()` reports every
    logical CPU of the machine (256 on a GPU box whose one-GPU job is granted 16): pools sized from it oversubscribe the quota 16x."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


def default_load_workers() -> int:
    """GEN:103 (`int(cpu_count * 0.8)`) on the CPUs this process may use, shared among the ranks of this node."""
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    return max(1, int(effective_cpus() * 0.8) // local_world)


def list_chunk_files(output_dir: Path) -> List[Path]:
    """All `*.json` under the tree except `._*` (GEN:97-98), SORTED: the order every rank agrees on."""
    return sorted(f for f in Path(output_dir).rglob("*.json") if not f.name.startswith("._"))


def load_chunk_files(files: Sequence[Path], min_quality: float = 0.8, num_workers: Optional[int] = None) -> List[List[Dict]]:
    """The kept chunks of each file, one list per file, in the order given.  `json.load` holds the GIL, so a corpus-sized list
    (GEN:103 sizes its pool for 70 k files) is parsed by a process pool — `spawn` context, as GEN:611-616 sets, which is also the
    only start method that is safe once the process has touched the GPU — fed contiguous slices through ordered `imap`; small
    lists are read by a thread pool in-process."""
    if num_workers is None:
        num_workers = default_load_workers()
    per_file: List[List[Dict]] = []
    if len(files) >= PROCESS_POOL_MIN_FILES and num_workers > 1:
        nproc = min(num_workers, 64, max(2, len(files) // 1024))
        step = 128
        jobs = [([str(f) for f in files[i:i + step]], min_quality) for i in range(0, len(files), step)]
        with mp.get_context("spawn").Pool(nproc) as pool:
            for part in pool.imap(_load_many_per_file, jobs, chunksize=1):
                per_file.extend(part)
    else:
        with ThreadPoolExecutor(max_workers=max(1, min(num_workers, 64))) as ex:
            per_file.extend(ex.map(lambda f: load_chunks_from_file(f, min_quality), files))
    return per_file


def _load_many_per_file(args: Tuple[List[str], float]) -> List[List[Dict]]:
    paths, min_quality = args
    return [load_chunks_from_file(Path(p), min_quality) for p in paths]


def load_chunks_parallel(output_dir: Path, min_quality: float = 0.8, num_workers: Optional[int] = None) -> List[Dict]:
    """All `*.json` under the tree except `._*` (GEN:94-129).  Files are visited in SORTED order and results concatenated in that
    order: the reference's imap_unordered order is not a contract, and a deterministic order is what lets N ranks agree on the
    shard boundaries."""
    files = list_chunk_files(output_dir)
    if num_workers is None:
        num_workers = default_load_workers()
    print(f"Loading chunks from {len(files):,} files using {num_workers} workers...")
    out: List[Dict] = []
    for part in load_chunk_files(files, min_quality, num_workers):
        if part:
            out.extend(part)
    print(f"Loaded {len(out):,} high-quality chunks (quality >= {min_quality})")
    return out


class RankChunks:
    """What ONE rank of an N-rank run holds of the corpus: the chunks of global rows [lo, hi) (its contiguous range of work quanta),
    `total` = the corpus' chunk count.  Built by `load_chunks_for_rank`: the tree is parsed ONCE across the ranks, not once per rank."""

    def __init__(self, chunks: List[Dict], lo: int, hi: int, total: int, files_total: int, files_parsed: int):
        self.chunks, self.lo, self.hi, self.total = chunks, lo, hi, total
        self.files_total, self.files_parsed = files_total, files_parsed


def load_chunks_for_rank(output_dir: Path, min_quality: float, num_workers: Optional[int], chunks_per_worker: int,
                         world: int, rank: int, gather_counts: Optional[Callable] = None) -> RankChunks:
    """Per-rank loading (the reference loads the tree once, with one pool: GEN:94-129; N ranks each loading ALL of it would be N
    parses and N copies of every chunk's text).  Rank r parses only its contiguous slice of the SORTED file list; an
    `all_gather_object` of the per-file kept-chunk counts gives every rank the global row offset of every file (the order contract —
    sorted files, chunks in file order — is unchanged); the rank's row range stays the range of work quanta the sharded dispatcher
    assigns (`shard_range` over `ceil(total / chunks_per_worker)` quanta, so every quantum is the same set of chunks as in a
    one-rank run), and the few files of a neighbouring slice that reach into that range are parsed here as well."""
    files = list_chunk_files(output_dir)
    f_lo, f_hi = shard_range(len(files), world, rank)
    if num_workers is None:
        num_workers = default_load_workers()
    print(f"[rank {rank}] Loading chunks from files [{f_lo:,}, {f_hi:,}) of {len(files):,} using {num_workers} workers...")
    mine = load_chunk_files(files[f_lo:f_hi], min_quality, num_workers)
    counts_local = [len(c) for c in mine]
    if gather_counts is None:
        import torch.distributed as dist
        def gather_counts(x):                                          # noqa: E306
            out: List = [None] * world
            dist.all_gather_object(out, x, group=host_group())
            return out
    counts: List[int] = [n for part in gather_counts(counts_local) for n in part]
    assert len(counts) == len(files), "ranks disagree on the file list"
    offsets = np.zeros(len(files) + 1, np.int64)
    np.cumsum(counts, out=offsets[1:])
    total = int(offsets[-1])
    n_quanta = (total + chunks_per_worker - 1) // chunks_per_worker
    q_lo, q_hi = shard_range(n_quanta, world, rank)
    lo, hi = min(total, q_lo * chunks_per_worker), min(total, q_hi * chunks_per_worker)
    chunks: List[Dict] = []
    extra = 0
    if hi > lo:
        fa = int(np.searchsorted(offsets, lo, side="right")) - 1       # first file holding row lo
        fb = int(np.searchsorted(offsets, hi, side="left"))            # one past the last file holding row hi - 1
        need = [i for i in range(fa, fb) if counts[i] > 0]
        outside = [i for i in need if not (f_lo <= i < f_hi)]
        extra = len(outside)
        fetched = dict(zip(outside, load_chunk_files([files[i] for i in outside], min_quality, num_workers))) if outside else {}
        for i in need:
            part = mine[i - f_lo] if f_lo <= i < f_hi else fetched[i]
            a, b = max(lo, int(offsets[i])) - int(offsets[i]), min(hi, int(offsets[i + 1])) - int(offsets[i])
            chunks.extend(part[a:b])
    assert len(chunks) == hi - lo, (len(chunks), lo, hi)
    print(f"[rank {rank}] parsed {f_hi - f_lo + extra} of {len(files)} files (own slice [{f_lo}, {f_hi}) + {extra} boundary files); "
          f"holds chunks [{lo:,}, {hi:,}) of {total:,} (quality >= {min_quality})")
    return RankChunks(chunks, lo, hi, total, len(files), f_hi - f_lo + extra)


# --------------------------------------------------------------------------------------------- embed
def generate_embeddings_worker(args: Tuple[List[str], str, int, int]) -> Tuple[int, List[np.ndarray], Optional[str]]:
    """One work quantum (<= chunks_per_worker texts): sub-batches of batch_size through model.encode;
    a failed sub-batch is retried per text, a failed text becomes a zero row; never raises (GEN:131-177)."""
    texts, model_name, batch_size, batch_idx = args
    try:
        model = get_worker_model(model_name)
        if model is None:
            return (batch_idx, [], "model not loaded")
        rows: List[np.ndarray] = []
        for s0 in range(0, len(texts), batch_size):
            sub = texts[s0:s0 + batch_size]
            try:
                rows.extend(model.encode(sub, batch_size=min(batch_size, len(sub)), normalize_embeddings=True,
                                         show_progress_bar=False, convert_to_numpy=True, convert_to_tensor=False))
            except Exception:
                print(f"[rank {os.environ.get('RANK', '0')}] sub-batch {s0 // batch_size} of quantum {batch_idx} failed, trying individual items")
                for t in sub:
                    try:
                        rows.extend(model.encode([t], normalize_embeddings=True, show_progress_bar=False, convert_to_numpy=True))
                    except Exception:
                        rows.append(np.zeros(model.get_sentence_embedding_dimension()))
        return (batch_idx, rows, None)
    except Exception as e:                                                   # noqa: BLE001
        msg = f"Error in rank {os.environ.get('RANK', '0')}: {e}"
        print(msg)
        traceback.print_exc()
        return (batch_idx, [], msg)


class ShardSink:
    """This rank's fp16 corpus shard, resident in HBM: rows [lo, hi) of the corpus, filled by the encoder kernels while the
    chunks are being embedded (`encode(..., device_f16_out=...)`) and searched where they lie — corpus rows never cross PCIe
    towards the device.  Only rows that came out of the reference's fallback policy (per-item retry / zero rows, GEN:155-169;
    host arrays by construction) are uploaded, a quantum at a time."""

    def __init__(self, model, lo: int, hi: int):
        import torch
        self.lo, self.hi = lo, hi
        self.rows = torch.zeros((hi - lo, model.get_sentence_embedding_dimension()), dtype=torch.float16,
                                device=model.encoder.device)

    def view(self, a: int, b: int):
        return self.rows[a - self.lo:b - self.lo]

    def put(self, a: int, rows: Sequence[np.ndarray]):
        import torch
        if len(rows):
            host = np.asarray(rows, dtype=np.float32).astype(np.float16)
            self.rows[a - self.lo:a - self.lo + len(rows)] = torch.from_numpy(host).to(self.rows.device)


def make_shard_sink(model, n_texts: int, chunks_per_worker: int, world: int, rank: int) -> Optional["ShardSink"]:
    """A sink for the row range this rank encodes (same split as the dispatchers below); None for a model with no device."""
    if getattr(model, "encoder", None) is None or not hasattr(model.encoder, "device"):
        return None
    n_quanta = (n_texts + chunks_per_worker - 1) // chunks_per_worker
    q_lo, q_hi = (0, n_quanta) if world == 1 else shard_range(n_quanta, world, rank)
    return ShardSink(model, min(n_texts, q_lo * chunks_per_worker), min(n_texts, q_hi * chunks_per_worker))


def _encode_quanta(texts: List[str], q_lo: int, q_hi: int, model_name: str, batch_size: int, chunks_per_worker: int,
                   super_quanta: int = 32, sink: Optional[ShardSink] = None, base: int = 0, total: Optional[int] = None):
    """Quanta [q_lo, q_hi) -> yields (qi, rows, err) exactly as `generate_embeddings_worker` would, but fast: a HIP sentence
    encoder gets `super_quanta` quanta per `encode()` call (rows do not depend on how texts are batched, so the result is the same;
    the tokenizer feeder and the 1024-sequence forwards then run at the device-bound rate instead of 200 texts at a time).  Any
    exception or row-count mismatch in a fast call falls back to the per-quantum worker, i.e. to the reference's policy
    (sub-batch retry, per-item retry, zero rows; GEN:131-177).  `texts` holds global rows [base, base + len(texts)) of a corpus of
    `total` chunks (per-rank loading); quanta are numbered globally."""
    total = len(texts) if total is None else total
    model = None
    try:
        model = get_worker_model(model_name)
    except Exception:
        model = None
    fast = model is not None and getattr(model, "coalesce_batches", False)
    qi = q_lo
    while qi < q_hi:
        qe = min(q_hi, qi + (super_quanta if fast else 1))
        a, b = qi * chunks_per_worker, min(total, qe * chunks_per_worker)
        rows = None
        if fast:
            try:
                extra = {"device_f16_out": sink.view(a, b)} if sink is not None else {}
                rows = model.encode(texts[a - base:b - base], batch_size=batch_size, normalize_embeddings=True, show_progress_bar=False,
                                    convert_to_numpy=True, convert_to_tensor=False, **extra)
                if len(rows) != b - a:
                    rows = None
            except Exception:
                rows = None
        for q in range(qi, qe):
            qa, qb = q * chunks_per_worker, min(total, (q + 1) * chunks_per_worker)
            if rows is not None:
                yield (q, list(rows[qa - a:qb - a]), None)
            else:
                res = generate_embeddings_worker((texts[qa - base:qb - base], model_name, batch_size, q))
                if sink is not None:                     # fallback rows are host arrays: this quantum's shard rows are replaced
                    try:
                        sink.view(qa, qb).zero_()
                        sink.put(qa, res[1][:qb - qa])
                    except Exception:                    # noqa: BLE001  (never raise per chunk; the rows stay zero)
                        pass
                yield res
        qi = qe


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


_host_pg = None


def init_distributed():
    """Bring up this rank's process group (one process per GPU; the reference's counterpart is the `mp.Pool` of GEN:197-217).
    The device is BOUND before the group exists: RCCL stages object collectives on `torch.cuda.current_device()`, which is cuda:0 in
    every rank until someone says otherwise — the first collective of the run (`load_chunks_for_rank`'s count exchange) runs before
    the model is loaded, and a communicator built with every rank on GPU 0 fails with "Duplicate GPU detected" or hangs
    (ADVICE r3).  `device_id` also makes the communicator eager, so a mis-launch fails here and not in the middle of the load."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    if torch.cuda.is_available():
        local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
        n_dev = torch.cuda.device_count()
        if local_rank >= n_dev:
            raise RuntimeError(f"LOCAL_RANK={local_rank} but only {n_dev} GPU(s) visible: launch one rank per GPU")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo")


def host_group():
    """Process group for the HOST-side object exchanges of an N-rank run (per-file chunk counts, metadata bookkeeping, chunk ids of
    merged hits): gloo over TCP, so that pickled Python objects never pass through device memory and the loader does not depend on
    the GPU at all.  The default group itself when that already is gloo (CPU runs / tests)."""
    global _host_pg
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        return None
    if _host_pg is None:
        _host_pg = dist.new_group(backend="gloo")
    return _host_pg


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def generate_embeddings_parallel(chunks: List[Dict], model_name: str = "all-mpnet-base-v2", batch_size: int = 200,
                                 num_workers: Optional[int] = None, chunks_per_worker: int = 500,
                                 sink: Optional[ShardSink] = None) -> List[np.ndarray]:
    """Embedding i <-> chunk i (GEN:179-269).  Quanta of chunks_per_worker texts; with torchrun each rank takes
    a contiguous range of quanta and the per-rank results are exchanged so every rank returns all N rows
    (missing quanta -> zero rows, count mismatch -> pad/truncate, as GEN:260-267)."""
    texts = [c["text"] for c in chunks]
    dist = _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    quanta = [(texts[i:i + chunks_per_worker], model_name, batch_size, qi)
              for qi, i in enumerate(range(0, len(texts), chunks_per_worker))]
    print(f"Generating embeddings for {len(texts):,} chunks on {world} GPU(s)...")
    print(f"Model: {model_name}, Batch size: {batch_size}, Chunks per worker: {chunks_per_worker}")
    q_lo, q_hi = shard_range(len(quanta), world, rank)
    done: Dict[int, List[np.ndarray]] = {}
    errors: List[str] = []
    for idx, rows, err in _encode_quanta(texts, q_lo, q_hi, model_name, batch_size, chunks_per_worker, sink=sink):
        if err:
            errors.append(f"Batch {idx}: {err}")
        if rows:
            done[idx] = rows
        else:
            print(f"Warning: Batch {idx} produced no embeddings")
    if dist and world > 1:
        gathered: List = [None] * world
        dist.all_gather_object(gathered, (done, errors), group=host_group())
        done, errors = {}, []
        for d, e in gathered:
            done.update(d); errors.extend(e)
    if errors:
        print(f"\n⚠️  {len(errors)} batches had errors:")
        for e in errors[:10]:
            print(f"  - {e}")
    embeddings: List[np.ndarray] = []
    missing = 0
    for qi in range(len(quanta)):
        rows = done.get(qi)
        if rows is not None:
            embeddings.extend(rows)
        else:
            missing += 1
    if missing:
        print(f"⚠️  {missing} batches missing, used zero vectors as fallback")
    print(f"Generated {len(embeddings):,} embeddings")
    if len(embeddings) != len(texts):
        print(f"⚠️  Warning: Embedding count ({len(embeddings)}) doesn't match text count ({len(texts)})")
        if len(embeddings) < len(texts):
            dim = len(embeddings[0]) if embeddings else 768
            embeddings.extend([np.zeros(dim)] * (len(texts) - len(embeddings)))
        else:
            embeddings = embeddings[:len(texts)]
    return embeddings


def generate_embeddings_sharded(chunks: List[Dict], model_name: str, batch_size: int = 200,
                                chunks_per_worker: int = 500, sink: Optional[ShardSink] = None,
                                span: Optional[Tuple[int, int, int]] = None) -> Tuple[np.ndarray, int, int]:
    """Multi-rank form of the dispatcher: this rank encodes a contiguous range of quanta and KEEPS its rows
    (no exchange of embeddings: at ~5 M chunks x 768 that would be 15 GB per rank for nothing).
    Returns (rows float32 [hi-lo, D], lo, hi) with row j <-> chunk lo + j; failed quanta are zero rows.
    `span = (lo, hi, total)`: `chunks` are only this rank's chunks, global rows [lo, hi) of `total` (`load_chunks_for_rank`);
    without it `chunks` is the whole corpus."""
    texts = [c["text"] for c in chunks]
    dist = _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    total = len(texts) if span is None else span[2]
    base = 0 if span is None else span[0]
    n_quanta = (total + chunks_per_worker - 1) // chunks_per_worker
    q_lo, q_hi = shard_range(n_quanta, world, rank)
    lo, hi = min(total, q_lo * chunks_per_worker), min(total, q_hi * chunks_per_worker)
    assert span is None or (span[0], span[1]) == (lo, hi), "the rank's chunks are not the rows its quanta cover"
    print(f"[rank {rank}] encoding chunks [{lo:,}, {hi:,}) of {total:,} ({q_hi - q_lo} quanta)")
    dim = get_worker_model(model_name).get_sentence_embedding_dimension()
    rows = np.zeros((hi - lo, dim), np.float32)
    errors = 0
    for idx, got, err in _encode_quanta(texts, q_lo, q_hi, model_name, batch_size, chunks_per_worker, sink=sink, base=base, total=total):
        a, b = idx * chunks_per_worker, min(total, (idx + 1) * chunks_per_worker)
        if err or len(got) != b - a:
            errors += 1
            print(f"Warning: Batch {idx} produced {len(got)} of {b - a} embeddings; missing rows stay zero")
        for j, e in enumerate(got[:b - a]):
            rows[a - lo + j] = e
    if errors:
        print(f"⚠️  [rank {rank}] {errors} batches had errors")
    return rows, lo, hi


def save_embeddings_sharded(chunks: List[Dict], rows: np.ndarray, lo: int, hi: int,
                            output_dir: str = "./embeddings_saved", out_dtype: str = "float64", prefetch=None,
                            total: Optional[int] = None):
    """Same three files as save_embeddings_to_disk_fallback, written cooperatively: rank 0 creates embeddings.npy
    (header + size), every rank stores its own row range through a memmap, rank 0 writes metadata/index.
    `total` given: `chunks` are only this rank's chunks (rows [lo, hi) of `total`); every rank then serialises ITS entries of
    metadata.json into a fragment file (beside the GPU work, when a `MetadataPrefetch` was started for it) and rank 0 joins the
    fragments in rank order — the same bytes `_dump_metadata` writes for the whole list."""
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    out = Path(output_dir)
    n, dim = (len(chunks) if total is None else total), rows.shape[1]
    if rank == 0:
        out.mkdir(parents=True, exist_ok=True)
        arr = np.lib.format.open_memmap(out / "embeddings.npy", mode="w+", dtype=np.dtype(out_dtype), shape=(n, dim))
        del arr
    if dist:
        dist.barrier()
    arr = np.load(out / "embeddings.npy", mmap_mode="r+")
    arr[lo:hi] = rows.astype(np.dtype(out_dtype))
    nbytes = arr.nbytes
    arr.flush()
    del arr
    if total is not None:                                   # this rank's entries of metadata.json
        frag = out / f"metadata.json.part{rank}"
        if prefetch is None or not prefetch.take(chunks, frag):
            _dump_metadata(chunks, frag, first_index=lo, fragment=True)
    if dist:
        dist.barrier()
    if rank == 0:
        if total is None:
            _write_metadata_and_index(chunks, out, n, dim, nbytes, prefetch)
        else:
            _join_metadata_fragments(out, world)
            print(f"✅ Saved metadata to {out / 'metadata.json'}")
            _write_index(out, n, dim, nbytes)
        print(f"✅ Saved {n:,} embeddings ({dim} dimensions) from {world} rank(s)")


def _join_metadata_fragments(out: Path, world: int):
    """metadata.json = "[\n" + the ranks' entry runs joined by ",\n" + "\n]" (what `_dump_metadata` writes for the whole list)."""
    with open(out / "metadata.json", "wb") as fh:
        first = True
        for r in range(world):
            frag = out / f"metadata.json.part{r}"
            if frag.stat().st_size:
                fh.write(b"[\n" if first else b",\n")
                first = False
                with open(frag, "rb") as src:
                    while True:
                        buf = src.read(1 << 24)
                        if not buf:
                            break
                        fh.write(buf)
            frag.unlink()
        fh.write(b"[]" if first else b"\n]")


# --------------------------------------------------------------------------------------------- write
def save_embeddings_to_disk_fallback(chunks: List[Dict], embeddings: Sequence, output_dir: str = "./embeddings_saved",
                                     out_dtype: str = "float64", prefetch=None):
    """embeddings.npy (float64 C-order [N, D] — the reference's `.tolist()` round trip yields float64),
    metadata.json (indent=2, ensure_ascii=False), index.json (GEN:271-321).  Streams through a memmap so a
    5 M x 768 corpus (30.7 GB) never needs a second in-RAM copy."""
    out = Path(output_dir)
    out.mkdir(parents=True, exist_ok=True)
    print(f"\nSaving embeddings to disk: {output_dir}")
    n = len(embeddings)
    dim = len(embeddings[0]) if n else 0
    arr = np.lib.format.open_memmap(out / "embeddings.npy", mode="w+", dtype=np.dtype(out_dtype), shape=(n, dim))
    step = 65536
    for s0 in range(0, n, step):
        arr[s0:s0 + step] = np.asarray(embeddings[s0:s0 + step], dtype=np.dtype(out_dtype))
    nbytes = arr.nbytes
    arr.flush()
    del arr
    print(f"✅ Saved embeddings to {out / 'embeddings.npy'}")
    _write_metadata_and_index(chunks, out, n, dim, nbytes, prefetch)
    print(f"✅ Saved {n:,} embeddings ({dim} dimensions)")
    print(f"   Total size: ~{nbytes / 1024 / 1024 / 1024:.2f} GB")


def _dump_metadata(chunks: List[Dict], path: Path, first_index: int = 0, fragment: bool = False):
    """metadata.json, byte for byte what `json.dump(metadata, f, indent=2, ensure_ascii=False)` writes at GEN:292-306, but
    streamed one chunk at a time: the reference builds the whole list and its serialisation in RAM (the text of ~5 M chunks
    twice over); here the peak is one entry.  `fragment`: only the entries (joined by ",\n", no brackets) of chunks that are
    rows first_index.. of the corpus — one rank's part of the file (`_join_metadata_fragments`)."""
    with open(path, "w", encoding="utf-8") as fh:
        if not chunks:
            fh.write("" if fragment else "[]")
            return
        if not fragment:
            fh.write("[\n")
        for i, ch in enumerate(chunks):
            m = ch.get("metadata", {})
            item = {"chunk_id": ch.get("chunk_id", f"chunk_{first_index + i}"), "paper_id": m.get("paper_id"),
                    "section": m.get("section"), "quality_score": m.get("quality_score"),
                    "text": ch["text"], "text_length": len(ch["text"])}
            body = json.dumps(item, indent=2, ensure_ascii=False)
            fh.write("  " + body.replace("\n", "\n  "))
            if i + 1 < len(chunks):
                fh.write(",\n")
            elif not fragment:
                fh.write("\n")
        if not fragment:
            fh.write("]")


class MetadataPrefetch(threading.Thread):
    """metadata.json depends on the chunk list only, so its (pure-Python, ~17 us per chunk) serialisation runs beside the
    GPU while the embeddings are being generated, into `<out>/metadata.json.partial`; the writer renames it into place at
    the point where the reference writes the file (GEN:300-311), or discards it when the run does not get that far."""

    def __init__(self, chunks: List[Dict], output_dir: str = "./embeddings_saved", first_index: int = 0,
                 fragment_of_rank: Optional[int] = None):
        super().__init__(daemon=True)
        self.chunks, self.out, self.ok = chunks, Path(output_dir), False
        self.first_index, self.fragment = first_index, fragment_of_rank is not None
        self.partial = self.out / ("metadata.json.partial" if fragment_of_rank is None else f"metadata.json.partial{fragment_of_rank}")

    def run(self):
        try:
            self.out.mkdir(parents=True, exist_ok=True)
            _dump_metadata(self.chunks, self.partial, first_index=self.first_index, fragment=self.fragment)
            self.ok = True
        except Exception:                                              # noqa: BLE001  (the writer then serialises itself)
            self.ok = False

    def take(self, chunks: List[Dict], dest: Path) -> bool:
        self.join()
        if self.ok and chunks is self.chunks and self.partial.exists() and dest.parent.resolve() == self.out.resolve():
            os.replace(self.partial, dest)
            return True
        self.discard()
        return False

    def discard(self):
        self.join()
        try:
            self.partial.unlink()
        except OSError:
            pass


def _write_metadata_and_index(chunks: List[Dict], out: Path, n: int, dim: int, nbytes: int, prefetch: Optional["MetadataPrefetch"] = None):
    if prefetch is None or not prefetch.take(chunks, out / "metadata.json"):
        _dump_metadata(chunks, out / "metadata.json")
    print(f"✅ Saved metadata to {out / 'metadata.json'}")
    _write_index(out, n, dim, nbytes)


def _write_index(out: Path, n: int, dim: int, nbytes: int):
    index = {"total_embeddings": n, "embedding_dimension": dim, "total_size_gb": nbytes / 1024 / 1024 / 1024}
    with open(out / "index.json", "w", encoding="utf-8") as fh:
        json.dump(index, fh, indent=2)


def store_in_chroma_batched(chunks: List[Dict], embeddings: Sequence, db_path: str = "./chroma_db",
                            collection_name: str = "scientific_papers", batch_size: int = 2000):
    """Vector-store hand-off (GEN:323-468): ids / documents / metadatas exactly as GEN:404-412, batches of
    `batch_size`, 3 tries per batch then per-item adds.  Raises ImportError when chromadb is absent."""
    import chromadb                                                    # noqa: F401  (lazy: optional dependency)
    n = min(len(chunks), len(embeddings))
    client = chromadb.PersistentClient(path=db_path)
    try:
        coll = client.get_collection(name=collection_name)
    except Exception:
        coll = client.create_collection(name=collection_name, metadata={"description": "Scientific paper chunks for RAG"})
    stored = 0
    for s0 in range(0, n, batch_size):
        ids, embs, docs, metas = [], [], [], []
        for j in range(s0, min(n, s0 + batch_size)):
            ch, m = chunks[j], chunks[j].get("metadata", {})
            ids.append(ch.get("chunk_id", f"chunk_{j}"))
            embs.append(np.asarray(embeddings[j]).tolist())
            docs.append(ch["text"])
            metas.append({"paper_id": str(m.get("paper_id", "unknown")), "section": str(m.get("section", "unknown")),
                          "quality_score": float(m.get("quality_score", 0.0)), "chunk_index": str(m.get("chunk_index", j))})
        for attempt in range(3):
            try:
                coll.add(ids=ids, embeddings=embs, documents=docs, metadatas=metas)
                stored += len(ids)
                break
            except Exception as e:                                     # noqa: BLE001
                if attempt < 2:
                    time.sleep(0.5)
                    continue
                print(f"Error storing batch {s0 // batch_size}: {e}")
                for k in range(len(ids)):
                    try:
                        coll.add(ids=[ids[k]], embeddings=[embs[k]], documents=[docs[k]], metadatas=[metas[k]])
                        stored += 1
                    except Exception:
                        pass
    if stored == 0:
        print("\n❌ Failed to store any embeddings in ChromaDB")
    else:
        print(f"✅ Stored {stored:,} embeddings in ChromaDB")
    return stored


# --------------------------------------------------------------------------------------------- search (added step)
def search_queries(model, chunks: List[Dict], shard: "ShardSink", queries: List[str], top_k: int = 10,
                   output_dir: str = "./embeddings_saved", chunk_base: int = 0) -> List[Dict]:
    """Brute-force cosine top-k (config.yaml:63-64 `top_k: 10`) over the rank's fp16 rows in HBM; with
    torchrun each rank holds the contiguous row shard it encoded and the partial top-k lists are
    all-gathered over RCCL and merged.  `shard` is the `ShardSink` the encode step filled: rows [lo, hi) are already where they
    will be searched and nothing is uploaded here; the queries are encoded straight into an fp16 device matrix.  (A consumer that
    only has the `.npy` rows on disk builds a `store.HipCollection` instead.)  `chunks[j]` is row `chunk_base + j` of the corpus:
    with per-rank loading a rank only knows the chunk ids of its own rows, so the ids of the merged hits are exchanged (one
    `all_gather_object` of at most Q x k small entries)."""
    import torch
    from .index import ShardIndex
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    dev = model.encoder.device
    qd = torch.empty((len(queries), model.get_sentence_embedding_dimension()), dtype=torch.float16, device=dev)
    model.encode(queries, batch_size=256, normalize_embeddings=True, convert_to_numpy=True, device_f16_out=qd, low_latency=True)
    # int8 pre-filter (same exact answers; +50 % shard memory, one quantisation pass): 1.5-1.7x the queries per second on small batches
    pre = "int8" if (shard.rows.shape[1] % 128 == 0 and shard.rows.shape[1] <= 1024 and shard.rows.shape[0] > 0) else None
    s, i = ShardIndex(shard.rows, idx_base=shard.lo, prefilter=pre, adaptive=True).search_distributed(qd, top_k)
    s, i = s.cpu().numpy(), i.cpu().numpy()
    names = {int(j): chunks[int(j) - chunk_base].get("chunk_id", f"chunk_{int(j)}")
             for j in np.unique(i) if chunk_base <= j < chunk_base + len(chunks)}
    if dist and world > 1:
        parts: List = [None] * world
        dist.all_gather_object(parts, names, group=host_group())
        names = {k: v for part in parts for k, v in part.items()}
    results = []
    for qi, text in enumerate(queries):
        hits = []
        for r in range(top_k):
            j = int(i[qi, r])
            if j < 0:
                continue
            hits.append({"rank": r + 1, "score": float(s[qi, r]), "index": j, "chunk_id": names.get(j, f"chunk_{j}")})
        results.append({"query": text, "results": hits})
    if rank == 0:
        with open(Path(output_dir) / "search_results.json", "w", encoding="utf-8") as fh:
            json.dump(results, fh, indent=2, ensure_ascii=False)
        print(f"✅ Saved top-{top_k} results for {len(queries)} queries to {Path(output_dir) / 'search_results.json'}")
    return results


# --------------------------------------------------------------------------------------------- main
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Generate embeddings on MI355X (drop-in for generate_embeddings_parallel.py)")
    p.add_argument("input_dir", type=str, help="Input directory (output_improved)")
    p.add_argument("--model", type=str, default="all-mpnet-base-v2",
                   help=f"Embedding model: one of {MODEL_CHOICES} (resolved to a LOCAL directory) or a model directory")
    p.add_argument("--min-quality", type=float, default=0.9, help="Minimum quality score (default: 0.9)")
    p.add_argument("--batch-size", type=int, default=200, help="Batch size for embedding generation (default: 200)")
    p.add_argument("--chroma-db", type=str, default="./chroma_db", help="Path to ChromaDB (default: ./chroma_db)")
    p.add_argument("--collection-name", type=str, default="scientific_papers", help="ChromaDB collection name")
    p.add_argument("--load-workers", type=int, default=None, help="Number of workers for loading (default: 80%% of CPU cores)")
    p.add_argument("--embedding-workers", type=int, default=None,
                   help="Accepted for compatibility; embedding parallelism = one process per GPU (torchrun)")
    p.add_argument("--store-batch-size", type=int, default=2000, help="Batch size for storing in ChromaDB (default: 2000)")
    p.add_argument("--chunks-per-worker", type=int, default=500, help="Chunks per work quantum (default: 500)")
    # additive
    p.add_argument("--model-dir", type=str, default=None, help="Directory holding local model folders (sets ARX_MODEL_DIR)")
    p.add_argument("--out-dtype", type=str, default="float64", choices=["float64", "float32", "float16"],
                   help="dtype of embeddings.npy (default float64 = the reference's layout)")
    p.add_argument("--queries", type=str, default=None, help="Text file, one query per line: run the cosine top-k step")
    p.add_argument("--top-k", type=int, default=10, help="Results per query (default: 10)")
    p.add_argument("--skip-chroma", action="store_true", help="Do not attempt ChromaDB ingestion")
    return p


def main(argv: Optional[Sequence[str]] = None, model_factory: Optional[Callable] = None) -> int:
    global _model, _model_name
    args = build_parser().parse_args(argv)
    input_dir = Path(args.input_dir)
    if not input_dir.exists():
        print(f"Error: Directory {input_dir} not found")
        return 1
    if args.model_dir:
        os.environ["ARX_MODEL_DIR"] = args.model_dir
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        init_distributed()
    cpu_count = effective_cpus()                                # usable by this job (cgroup quota), not the machine's logical CPUs
    print("=" * 80)
    print("PARALLEL EMBEDDING GENERATION - MI355X")
    print("=" * 80)
    print(f"\nSystem: {cpu_count} CPU cores, {world} GPU rank(s)")
    print("Configuration:")
    print(f"  Model: {args.model}")
    print(f"  Min quality: {args.min_quality}")
    print(f"  Batch size: {args.batch_size}")
    print(f"  Load workers: {args.load_workers or default_load_workers()}")
    print(f"  Store batch size: {args.store_batch_size}")
    print(f"  ChromaDB: {args.chroma_db}\n")
    meta_prefetch = None
    try:
        t_start = time.time()
        span = None
        if world > 1:
            # per-rank loading: every rank parses its slice of the sorted file list and ends up holding only the chunks it encodes
            rc_ = load_chunks_for_rank(input_dir, args.min_quality, args.load_workers, args.chunks_per_worker, world, rank)
            chunks, span, n_chunks = rc_.chunks, (rc_.lo, rc_.hi, rc_.total), rc_.total
        else:
            chunks = load_chunks_parallel(input_dir, min_quality=args.min_quality, num_workers=args.load_workers)
            n_chunks = len(chunks)
        if not n_chunks:
            print("No chunks found!")
            return 1
        load_time = time.time() - t_start
        print(f"Loading completed in {load_time:.1f} seconds\n")
        try:
            _model, _model_name = None, None
            init_worker_model(args.model, model_factory)
        except (ImportError, FileNotFoundError, OSError, RuntimeError) as e:
            print(f"Error: embedding backend not available: {e}")
            return 1
        # metadata.json is serialised beside the GPU work (this rank's entries when the chunks are per rank); renamed into place by the writer
        if span is not None:
            meta_prefetch = MetadataPrefetch(chunks, "./embeddings_saved", first_index=span[0], fragment_of_rank=rank)
            meta_prefetch.start()
        elif rank == 0:
            meta_prefetch = MetadataPrefetch(chunks, "./embeddings_saved")
            meta_prefetch.start()
        t0 = time.time()
        qs = []
        if args.queries:
            qs = [ln.strip() for ln in Path(args.queries).read_text(encoding="utf-8").splitlines() if ln.strip()]
        # the search step works on the rows where the encoder leaves them: an fp16 shard in this rank's HBM
        sink = make_shard_sink(_model, n_chunks, args.chunks_per_worker, world, rank) if qs else None
        if world > 1:
            # one process per GPU: every rank encodes, keeps and writes its own contiguous row range
            embeddings, lo, hi = generate_embeddings_sharded(chunks, args.model, args.batch_size, args.chunks_per_worker, sink=sink, span=span)
        else:
            embeddings = generate_embeddings_parallel(chunks, model_name=args.model, batch_size=args.batch_size,
                                                      num_workers=args.embedding_workers, chunks_per_worker=args.chunks_per_worker,
                                                      sink=sink)
        embedding_time = time.time() - t0
        print(f"Embedding generation completed in {embedding_time:.1f} seconds ({embedding_time / 60:.1f} min)\n")
        if world > 1:
            save_embeddings_sharded(chunks, embeddings, lo, hi, output_dir="./embeddings_saved", out_dtype=args.out_dtype,
                                    prefetch=meta_prefetch, total=n_chunks)
        elif rank == 0:
            print("Saving embeddings to disk as backup...")
            save_embeddings_to_disk_fallback(chunks, embeddings, output_dir="./embeddings_saved", out_dtype=args.out_dtype,
                                             prefetch=meta_prefetch)
            print()
        meta_prefetch = None
        if qs:
            if sink is None:
                print("⚠️  --queries needs the HIP encoder (the search step runs over the shard it leaves in HBM): skipped")
            else:
                search_queries(_model, chunks, sink, qs, top_k=args.top_k, chunk_base=span[0] if span else 0)
        store_time = 0.0
        if rank == 0 and not args.skip_chroma:
            try:
                import chromadb                                        # noqa: F401
            except ImportError:
                print("Error: chromadb not available")
                print("Install with: pip install chromadb")
                return 1
            t0 = time.time()
            if world > 1:                       # rank 0 only holds its shard: ingest from the files all ranks just wrote
                embeddings = np.load(Path("./embeddings_saved") / "embeddings.npy", mmap_mode="r")
                chunks = load_chunks_parallel(input_dir, min_quality=args.min_quality, num_workers=args.load_workers)
            try:
                store_in_chroma_batched(chunks, embeddings, db_path=args.chroma_db, collection_name=args.collection_name,
                                        batch_size=args.store_batch_size)
                store_time = time.time() - t0
                print(f"Storage completed in {store_time:.1f} seconds")
            except Exception as e:                                     # noqa: BLE001
                store_time = time.time() - t0
                print(f"⚠️  ChromaDB storage failed after {store_time:.1f} seconds: {e}")
                print("✅ Embeddings are safely saved to ./embeddings_saved/")
        total = time.time() - t_start
        print("\n" + "=" * 80)
        print("EMBEDDING GENERATION COMPLETE")
        print("=" * 80)
        print(f"Chunks processed: {n_chunks:,}")
        print(f"Embeddings generated: {n_chunks if world > 1 else len(embeddings):,}")
        if len(embeddings):
            print(f"Embedding dimensions: {len(embeddings[0])}")
        print(f"Stored in: {args.chroma_db}\n")
        print("Timing:")
        print(f"  Loading: {load_time:.1f}s")
        print(f"  Embedding: {embedding_time:.1f}s ({embedding_time / 60:.1f} min)")
        print(f"  Storage: {store_time:.1f}s")
        print(f"  Total: {total:.1f}s ({total / 60:.1f} min)")
        print("=" * 80)
        return 0
    except KeyboardInterrupt:
        print("\n\n⚠️  Process interrupted by user")
        return 1
    except Exception as e:                                             # noqa: BLE001
        print(f"\n\n❌ Fatal error: {e}")
        traceback.print_exc()
        return 1
    finally:
        if meta_prefetch is not None:            # the run ended before the writer took the file: nothing half-written stays behind
            meta_prefetch.discard()


if __name__ == "__main__":
    sys.exit(main())
