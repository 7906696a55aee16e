"""HBM-resident brute-force cosine top-k (`retrieval.top_k: 10`, 3-chunks/pipeline/config.yaml:63-64).

One `ShardIndex` per process/GPU holds that rank's rows of the fp16 corpus (the rows it encoded; they
never move).  `search` = local exact top-k through the C ABI; `search_distributed` adds the single
exchange step of the path: an all-gather of the per-shard [Q, k] partial results over RCCL, then a
merge kernel (ties -> lower global row id).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib


class ShardIndex:
    def __init__(self, corpus_f16: torch.Tensor, idx_base: int = 0, prefilter: Optional[str] = None, adaptive: bool = False):
        """`prefilter="int8"` (dim % 128 == 0, <= 1024): also keep an int8 representation of the rows (+50 % memory) and run the first
        pass of every search over it — the same exact top-k (`arx_topk_search_i8`), 1.4-1.7x the queries per second on rows that
        quantise well.  `adaptive=True` (the CLI and `HipCollection` pass it): the index reads the certificate counters after its
        first searches (a 16-byte copy + stream sync each) and switches the pre-filter OFF for good when more than a quarter of a
        batch's queries overflowed their candidate lists — clustered / outlier-heavy rows on which the int8 bound is too slack to
        pay (answers are exact either way; this only picks the faster first pass)."""
        assert corpus_f16.is_cuda and corpus_f16.dtype == torch.float16 and corpus_f16.dim() == 2
        assert corpus_f16.stride(1) == 1 and corpus_f16.stride(0) == corpus_f16.shape[1], "corpus must be dense row-major"
        assert prefilter in (None, "int8")
        self.lib = _lib.load()
        self.corpus = corpus_f16
        self.n_rows, self.dim = corpus_f16.shape
        self.idx_base = int(idx_base)
        self._ws: Optional[torch.Tensor] = None
        self._i8: Optional[torch.Tensor] = None
        self._i8_version = -1
        self._adaptive, self._i8_searches, self.prefilter_disabled = bool(adaptive), 0, False
        if prefilter == "int8" and self.n_rows > 0:
            self.build_int8()

    def build_int8(self):
        """(Re)build the int8 pre-filter from the current fp16 rows.  `search` calls it by itself when the corpus tensor was written
        to since the last build (torch's version counter): int8 values of rows that no longer exist are not upper bounds of anything,
        and a certificate computed from them could certify a wrong answer."""
        need = self.lib.arx_topk_i8_index_bytes(self.n_rows, self.dim)
        if need < 0:
            raise _lib.ArxError(f"int8 pre-filter needs dim % 128 == 0 and dim <= 1024 (dim={self.dim})")
        if self._i8 is None or self._i8.numel() != need:
            self._i8 = torch.empty(need, dtype=torch.uint8, device=self.corpus.device)
        _lib.check(self.lib.arx_topk_build_i8(self.corpus.data_ptr(), self.n_rows, self.dim, self._i8.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "arx_topk_build_i8")
        self._i8_version = self.corpus._version

    def _workspace(self, nq: int, k: int) -> torch.Tensor:
        need = self.lib.arx_topk_workspace_bytes(self.n_rows, nq, self.dim, k)
        if need < 0:
            raise _lib.ArxError(f"unsupported search shape n_rows={self.n_rows} nq={nq} dim={self.dim} k={k}")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.corpus.device)
        return self._ws

    def search_many(self, batches, k: int = 10, distributed: bool = False, group=None):
        """A STREAM of query batches, two in flight: batch b + 1's scan of the shard (pass A: every CU, HBM-bound) runs while batch b's
        select / rescore / certificate tail (a few dozen blocks, latency-bound) and — under a process group — its all-gather of the
        [Q, k] partials and the merge finish (SURVEY.md §8e "Overlap": on a 625 k-row shard the tail and the exchange are as long as the
        pass itself).  Two side streams, each with its own workspace; the caller's stream waits for both before this returns.
        Same results as calling `search` / `search_distributed` batch by batch."""
        import torch.distributed as dist
        dev = self.corpus.device
        cur = torch.cuda.current_stream(dev)
        if not hasattr(self, "_lanes"):
            self._lanes = [(torch.cuda.Stream(dev), None), (torch.cuda.Stream(dev), None)]
        ready = torch.cuda.Event()
        ready.record(cur)
        out = []
        keep_ws = self._ws
        for b, q in enumerate(batches):
            st, ws = self._lanes[b & 1]
            st.wait_event(ready)
            with torch.cuda.stream(st):
                self._ws = ws
                s, i = self.search(q, k)
                self._lanes[b & 1] = (st, self._ws)
                if distributed and dist.is_initialized():
                    s, i = merge_partials(*gather_partials(s, i, group), k)
                q.record_stream(st); s.record_stream(cur); i.record_stream(cur)
            out.append((s, i))
        self._ws = keep_ws
        for st, _ in self._lanes:
            cur.wait_stream(st)
        return out

    def search(self, queries_f16: torch.Tensor, k: int = 10) -> Tuple[torch.Tensor, torch.Tensor]:
        """queries fp16 [Q, D] (device) -> (scores f32 [Q, k], ids int64 [Q, k]); ids are global
        (local row + idx_base); score desc, ties -> lower id; (-inf, -1) pads when k > n_rows."""
        q = queries_f16
        assert q.is_cuda and q.dtype == torch.float16 and q.dim() == 2 and q.shape[1] == self.dim and q.is_contiguous()
        nq = q.shape[0]
        scores = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        if nq == 0:
            return scores, ids
        if self.n_rows == 0:
            scores.fill_(float("-inf")); ids.fill_(-1)
            return scores, ids
        ws = self._workspace(nq, k)
        use_i8 = self._i8 is not None and not self.prefilter_disabled
        if use_i8 and self.corpus._version != self._i8_version:      # rows written since the int8 copy was made (ShardSink.put, the encoder)
            self.build_int8()
        if use_i8:
            rc = self.lib.arx_topk_search_i8(self.corpus.data_ptr(), self._i8.data_ptr(), self.n_rows, q.data_ptr(), nq, self.dim, k,
                                             scores.data_ptr(), ids.data_ptr(), self.idx_base, ws.data_ptr(), ws.numel(),
                                             torch.cuda.current_stream().cuda_stream)
        else:
            rc = self.lib.arx_topk_search(self.corpus.data_ptr(), self.n_rows, q.data_ptr(), nq, self.dim, k,
                                          scores.data_ptr(), ids.data_ptr(), self.idx_base, ws.data_ptr(), ws.numel(),
                                          torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "arx_topk_search")
        if use_i8 and self._adaptive and nq <= 1024 and (self._i8_searches < 4 or self._i8_searches % 64 == 0):
            flagged, _ = self.certificate_stats()
            if flagged * 4 > nq:
                import sys
                print(f"[arx] int8 pre-filter switched off for this index: {flagged} of {nq} queries overflowed their candidate lists "
                      f"(rows on which the int8 bound is too slack); the fp16 pass answers from here on", file=sys.stderr)
                self.prefilter_disabled = True
        self._i8_searches += 1 if use_i8 else 0
        return scores, ids

    def certificate_stats(self) -> Tuple[int, int]:
        """(queries whose first selection could not be certified, extra 64-row groups rescored for them) of the LAST `search`
        on this index: the answer is exact either way (csrc/search.hip, rescore_kernel step 5); the counters say how often the
        slow path ran.  Where the int8 pre-filter ran the pair is (queries whose OWN candidate lists overflowed and went to the
        exhaustive kernel, (query, group) candidates the pre-filter's bounds let through) instead.
        Synchronises on the current stream."""
        import ctypes as C
        if self._ws is None:
            return (0, 0)
        a, b = C.c_int64(0), C.c_int64(0)
        _lib.check(self.lib.arx_topk_stats(self._ws.data_ptr(), C.byref(a), C.byref(b), torch.cuda.current_stream().cuda_stream),
                   "arx_topk_stats")
        return (int(a.value), int(b.value))

    def search_distributed(self, queries_f16: torch.Tensor, k: int = 10, group=None):
        """Every rank passes the SAME queries; returns the global top-k on every rank."""
        import torch.distributed as dist
        s, i = self.search(queries_f16, k)
        if not dist.is_initialized():
            return s, i
        # world size 1 takes the same gather + merge path (a 1-part merge is the identity): one code path to test
        all_s, all_i = gather_partials(s, i, group)
        return merge_partials(all_s, all_i, k)


def gather_partials(s: torch.Tensor, i: torch.Tensor, group=None):
    """The path's only exchange step: every rank contributes its [Q, k] partial (scores, global ids);
    returns the stacked [P, Q, k] tensors on every rank (RCCL on GPU tensors, gloo on CPU tensors)."""
    import torch.distributed as dist
    P = dist.get_world_size(group)
    all_s = torch.empty((P,) + tuple(s.shape), dtype=s.dtype, device=s.device)
    all_i = torch.empty((P,) + tuple(i.shape), dtype=i.dtype, device=i.device)
    if s.is_cuda:
        dist.all_gather_into_tensor(all_s, s.contiguous(), group=group)
        dist.all_gather_into_tensor(all_i, i.contiguous(), group=group)
    else:                                   # gloo: list form
        ls = [torch.empty_like(s) for _ in range(P)]; li = [torch.empty_like(i) for _ in range(P)]
        dist.all_gather(ls, s.contiguous(), group=group); dist.all_gather(li, i.contiguous(), group=group)
        all_s = torch.stack(ls); all_i = torch.stack(li)
    return all_s, all_i


def merge_partials(all_scores: torch.Tensor, all_ids: torch.Tensor, k: int):
    """[P, Q, k] partial lists (device) -> merged [Q, k] via the HIP merge kernel."""
    lib = _lib.load()
    P, nq, kk = all_scores.shape
    assert kk == k and all_scores.is_contiguous() and all_ids.is_contiguous()
    out_s = torch.empty((nq, k), dtype=torch.float32, device=all_scores.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=all_scores.device)
    rc = lib.arx_topk_merge(all_scores.data_ptr(), all_ids.data_ptr(), P, nq, k, out_s.data_ptr(), out_i.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "arx_topk_merge")
    return out_s, out_i


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """rows [r*ceil(N/P), min(N, (r+1)*ceil(N/P)))  (SURVEY.md §8e)."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


def fill_unit_rows(n_rows: int, dim: int, seed: int, device="cuda:0", row_base: int = 0) -> torch.Tensor:
    """Synthetic corpus/queries generated directly in HBM (bench cfg 3).  `row_base`: this tensor is rows [row_base, row_base + n_rows)
    of the corpus the seed defines — every rank of a sharded run fills its own slice of ONE corpus (bench cfg 4)."""
    lib = _lib.load()
    t = torch.empty((n_rows, dim), dtype=torch.float16, device=device)
    if n_rows > 0:
        _lib.check(lib.arx_fill_unit_rows_f16_at(t.data_ptr(), n_rows, dim, seed, row_base, torch.cuda.current_stream(t.device).cuda_stream),
                   "arx_fill_unit_rows_f16_at")
    return t
