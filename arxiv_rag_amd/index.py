"""HBM-resident brute-force cosine top-k (`retrieval.top_k: 10`, 3-chunks/pipeline/config.yaml:63-64).

One `ShardIndex` per process/GPU holds that rank's rows of the fp16 corpus (the rows it encoded; they
never move).  `search` = local exact top-k through the C ABI; `search_distributed` adds the single
exchange step of the path: an all-gather of the per-shard [Q, k] partial results over RCCL, then a
merge kernel (ties -> lower global row id).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib


class ShardIndex:
    def __init__(self, corpus_f16: torch.Tensor, idx_base: int = 0, prefilter: Optional[str] = None, adaptive: bool = False,
                 i8_max_queries: Optional[int] = None, max_row_norm: Optional[float] = None, centre_query: Optional[bool] = None):
        """`prefilter="int8"` (dim % 128 == 0, <= 1024): also keep an int8 representation of the rows (+50 % memory) and run the first
        pass of every search over it — the same exact top-k (`arx_topk_search_i8`), 1.4-1.8x the queries per second on rows that
        quantise well (the index quantises rows minus the shard's mean and, `centre_query`, queries minus their component along it, so rows
        sharing a common component of any size still do; profiles/r04/int8_centred_index.md).  `adaptive=True` (the CLI and `HipCollection` pass it): the index reads the certificate counters after its
        first searches (a 16-byte copy + stream sync each) and switches the pre-filter OFF for good when more than a quarter of a
        batch's queries overflowed their candidate lists — clustered / outlier-heavy rows on which the int8 bound is too slack to
        pay (answers are exact either way; this only picks the faster first pass).
        `i8_max_queries`: THIS index's crossover (query batches above it take the fp16 first pass; None = the library default, 0 = never
        int8) — a per-index policy passed with every call, not a library setting.
        `centre_query`: the int8 pass also quantises the QUERY minus its component along the shard's mean direction (ARX_TOPK_I8_CENTRE_QUERY: one
        more instruction per value in pass A's epilogue, same exact answers) — what keeps the pre-filter useful on strongly anisotropic rows
        (mean pairwise cosine beyond ~0.75).  None (default) = decided from the built index: on when |mean|^2 >= 0.25.
        `max_row_norm`: a bound on the rows' L2 norms the caller vouches for; None (default) = measured on the device
        (`arx_rows_max_norm_f16`, again whenever the tensor was written to) — the exactness certificate's tolerance scales with it, so
        the answers are the exact top-k of the dot products for rows of ANY norm, cosine when they are unit."""
        assert corpus_f16.is_cuda and corpus_f16.dtype == torch.float16 and corpus_f16.dim() == 2
        assert corpus_f16.stride(1) == 1 and corpus_f16.stride(0) == corpus_f16.shape[1], "corpus must be dense row-major"
        assert prefilter in (None, "int8")
        self.lib = _lib.load()
        self.corpus = corpus_f16
        self.n_rows, self.dim = corpus_f16.shape
        self.idx_base = int(idx_base)
        self._ws: Optional[torch.Tensor] = None            # workspace of plain `search` calls (and of the LAST call: certificate_stats)
        self._last_ws: Optional[torch.Tensor] = None
        self._i8: Optional[torch.Tensor] = None
        self._i8_version = -1
        self._adaptive, self._i8_searches, self.prefilter_disabled = bool(adaptive), 0, False
        self.i8_max_queries = i8_max_queries
        self._norm_given = None if max_row_norm is None else float(max_row_norm)
        self._norm, self._norm_version = None, -1
        self._pipe = None                                  # search_many's streams / workspaces
        self._centre_query_given, self.centre_query, self.i8_mean_norm = centre_query, bool(centre_query), 0.0
        if prefilter == "int8" and self.n_rows > 0:
            self.build_int8()

    # ---- state derived from the rows (refreshed when the tensor was written to) -------------------------------------------
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
        info = (C.c_float * 3)()
        _lib.check(self.lib.arx_topk_i8_index_info(self._i8.data_ptr(), self.n_rows, self.dim, info, torch.cuda.current_stream().cuda_stream),
                   "arx_topk_i8_index_info")
        self.i8_mean_norm = float(info[0])
        self.centre_query = (self.i8_mean_norm ** 2 >= 0.25) if self._centre_query_given is None else bool(self._centre_query_given)

    def max_row_norm(self) -> float:
        """Upper bound on the L2 norm of the shard's rows (measured on the device unless the caller gave one)."""
        if self._norm_given is not None:
            return self._norm_given
        if self._norm is None or self._norm_version != self.corpus._version or self._norm_version < 0:
            out = torch.empty(1, dtype=torch.float32, device=self.corpus.device)
            _lib.check(self.lib.arx_rows_max_norm_f16(self.corpus.data_ptr(), self.n_rows, self.dim, out.data_ptr(),
                                                      torch.cuda.current_stream().cuda_stream), "arx_rows_max_norm_f16")
            v = float(out.item())
            if not (v == v and v < float("inf")):
                raise _lib.ArxError("the shard holds a non-finite row (inf / NaN): top-k over it is undefined")
            # rows at most the unit bound keep the library's default tolerance (1 + 2^-9); an all-zero shard still needs a positive bound
            self._norm, self._norm_version = max(v, 1e-30), self.corpus._version
        return self._norm

    def _refresh(self):
        """Bring the int8 copy and the norm bound up to date with the rows ON THE CURRENT STREAM, before any search kernel that reads
        them is enqueued on any stream (search_many dispatches to side streams: ADVICE r3)."""
        if self.n_rows == 0:
            return
        self.max_row_norm()
        if self._i8 is not None and not self.prefilter_disabled and self.corpus._version != self._i8_version:
            self.build_int8()

    def invalidate(self):
        """Tell the index its rows were written BEHIND torch's back (through a raw pointer: the encoder kernels filling a slice of the shard
        via `device_f16_out`, another library): the tensor's version counter does not see such writes, so the int8 copy and the norm
        bound would silently describe rows that no longer exist.  The next search re-derives both.  (Writes through torch ops are noticed
        without this.)"""
        self._norm_version = -1
        self._i8_version = -1

    def _use_i8(self) -> bool:
        return self._i8 is not None and not self.prefilter_disabled and self.i8_max_queries != 0

    def _options(self, cu_limit: int = 0, flags: int = 0, **debug) -> "_lib.TopkOptionsC":
        o = _lib.TopkOptionsC()
        o.i8_max_queries = 0 if self.i8_max_queries is None else (-1 if self.i8_max_queries <= 0 else int(self.i8_max_queries))
        o.max_row_norm = max(self.max_row_norm(), 1.0 + 1.0 / 512.0)      # never below the encoder's unit-row bound
        o.cu_limit, o.flags = int(cu_limit), int(flags) | (_lib.TOPK_I8_CENTRE_QUERY if (self.centre_query and self._use_i8()) else 0)
        o.debug_tau_mult = float(debug.get("tau_mult", 0.0))
        o.debug_drop_best = int(debug.get("drop_best", 0))
        return o

    def workspace_bytes(self, nq: int, k: int) -> int:
        """Bytes of workspace THIS index's searches of nq queries need (the int8 pass needs the larger layout)."""
        fn = self.lib.arx_topk_workspace_bytes_i8 if self._use_i8() else self.lib.arx_topk_workspace_bytes
        need = fn(self.n_rows, nq, self.dim, k)
        if need < 0:
            raise _lib.ArxError(f"unsupported search shape n_rows={self.n_rows} nq={nq} dim={self.dim} k={k}")
        return int(need)

    def alloc_workspace(self, nq: int, k: int) -> torch.Tensor:
        return torch.empty(self.workspace_bytes(nq, k), dtype=torch.uint8, device=self.corpus.device)

    def _workspace(self, nq: int, k: int) -> torch.Tensor:
        need = self.workspace_bytes(nq, k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.corpus.device)
        return self._ws

    # ---- one batch ------------------------------------------------------------------------------------------------------
    def search(self, queries_f16: torch.Tensor, k: int = 10, ws: Optional[torch.Tensor] = None, out=None,
               _opt: Optional["_lib.TopkOptionsC"] = None, _stream: Optional[int] = None, **debug) -> Tuple[torch.Tensor, torch.Tensor]:
        """queries fp16 [Q, D] (device) -> (scores f32 [Q, k], ids int64 [Q, k]); ids are global
        (local row + idx_base); score desc, ties -> lower id; (-inf, -1) pads when k > n_rows.
        `ws`: the device workspace this call uses (default: the index's own, which makes the index one-caller-at-a-time; callers that
        search one index from several host threads or streams pass their own, `alloc_workspace`).  `out = (scores, ids)` preallocated.
        `tau_mult` / `drop_best`: the certificate's test hooks, `flags`: ARX_TOPK_* (include/arx.h) — all per call."""
        q = queries_f16
        assert q.is_cuda and q.dtype == torch.float16 and q.dim() == 2 and q.shape[1] == self.dim and q.is_contiguous()
        nq = q.shape[0]
        if out is None:
            scores = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        else:
            scores, ids = out
            assert scores.shape == (nq, k) and ids.shape == (nq, k) and scores.is_contiguous() and ids.is_contiguous()
        if nq == 0:
            return scores, ids
        if self.n_rows == 0:
            scores.fill_(float("-inf")); ids.fill_(-1)
            return scores, ids
        if _opt is None:
            self._refresh()                                  # rows written since the int8 copy / norm bound were made (ShardSink.put, the encoder)
            _opt = self._options(flags=debug.pop("flags", 0), **debug)
        if ws is None:
            ws = self._workspace(nq, k)
        use_i8 = self._use_i8()
        st = torch.cuda.current_stream().cuda_stream if _stream is None else _stream
        rc = self.lib.arx_topk_search_opt(self.corpus.data_ptr(), self._i8.data_ptr() if use_i8 else None, self.n_rows, q.data_ptr(), nq,
                                          self.dim, k, scores.data_ptr(), ids.data_ptr(), self.idx_base, ws.data_ptr(), ws.numel(),
                                          C.byref(_opt), st)
        _lib.check(rc, "arx_topk_search_opt")
        self._last_ws = ws
        if _opt.flags & (_lib.TOPK_SCAN_ONLY | _lib.TOPK_TAIL_ONLY):
            return scores, ids
        if use_i8 and self._adaptive and nq <= 1024 and (self._i8_searches < 4 or self._i8_searches % 64 == 0):
            flagged, _ = self.certificate_stats()
            if flagged * 4 > nq:
                import sys
                print(f"[arx] int8 pre-filter switched off for this index: {flagged} of {nq} queries overflowed their candidate lists "
                      f"(rows on which the int8 bound is too slack); the fp16 pass answers from here on", file=sys.stderr)
                self.prefilter_disabled = True
        self._i8_searches += 1 if use_i8 else 0
        return scores, ids

    def certificate_stats(self, ws: Optional[torch.Tensor] = None) -> Tuple[int, int]:
        """(queries whose first selection could not be certified, extra 64-row groups rescored for them) of the LAST `search`
        on this index (or on `ws`): the answer is exact either way (csrc/search_tail.h, rescore_kernel step 5); the counters say how often
        the slow path ran.  Where the int8 pre-filter ran the pair is (queries whose OWN candidate lists overflowed and went to the
        exhaustive kernel, (query, group) candidates the pre-filter's bounds let through) instead.
        Synchronises on the current stream."""
        ws = self._last_ws if ws is None else ws
        if ws is None:
            return (0, 0)
        a, b = C.c_int64(0), C.c_int64(0)
        _lib.check(self.lib.arx_topk_stats(ws.data_ptr(), C.byref(a), C.byref(b), torch.cuda.current_stream().cuda_stream),
                   "arx_topk_stats")
        return (int(a.value), int(b.value))

    # ---- a stream of batches --------------------------------------------------------------------------------------------
    def _pipeline(self, tail_cus: int, lanes: int):
        """`lanes` (scan, tail) stream pairs for `tail_cus` (process-wide per device, see `_cu_streams`) + one workspace per lane."""
        if self._pipe is None or (self._pipe["tail_cus"], len(self._pipe["lanes"])) != (tail_cus, lanes):
            self._pipe = {"tail_cus": tail_cus, "lanes": [_cu_streams(self.corpus.device, tail_cus, ln) for ln in range(lanes)],
                          "ws": [None] * lanes, "xchg": torch.cuda.Stream(self.corpus.device)}
        return self._pipe

    def search_many(self, batches, k: int = 10, distributed: bool = False, group=None, tail_cus: int = 0, lanes: int = 2):
        """A STREAM of query batches (each <= 1 024 queries), pipelined over `lanes` pairs of HIP streams: a lane's `scan` stream (pass A:
        HBM-bound) and its `tail` stream (select / exact rescoring / certificate), batch b on lane b % lanes, one workspace per lane.
        While one lane is in its tail — or in the few microseconds of dependent-launch latency between its kernels — the other lane's
        scan has the memory system to itself (SURVEY.md §8e "Overlap": on a 625 k-row shard the tail and the exchange are as long as the
        pass itself).  Under a process group batch b's all-gather of the [Q, k] partials and the merge run on a third, ordinary stream.
        `tail_cus` > 0 gives the two streams of a lane DISJOINT compute-unit masks (scan on all but `tail_cus` CUs, tail on the rest:
        `arx_stream_create_cu_mask`), so that the tail's blocks never wait for scan blocks to drain.  Measured on the 625 k x 768 shard
        (profiles/r04/search_many_lanes_cu_split.md): it LOSES — 0.191 ms per 64-query batch with plain streams, 0.22-0.28 with 16-64
        tail CUs: once the tail reads one row per selected group (round 4) it is short enough that taking CUs from the scan costs more
        than the waiting did — hence the default 0.
        The caller's stream waits for every stream before this returns.  Same results, bit for bit, as `search` /
        `search_distributed` batch by batch."""
        import torch.distributed as dist
        batches = list(batches)
        if not batches:
            return []
        dev = self.corpus.device
        cur = torch.cuda.current_stream(dev)
        if self.n_rows == 0 or any(q.shape[0] > 1024 or q.shape[0] == 0 for q in batches):
            return [self.search_distributed(q, k, group) if distributed else self.search(q, k) for q in batches]
        self._refresh()                                      # on the caller's stream, BEFORE any lane may read the int8 copy / norm bound
        lanes = max(1, min(int(lanes), len(batches)))
        pipe = self._pipeline(tail_cus, lanes)
        need = max(self.workspace_bytes(nq_b, k) for nq_b in {q.shape[0] for q in batches})      # (not monotonic in the batch size: small batches carry aux words)
        for j in range(lanes):
            if pipe["ws"][j] is None or pipe["ws"][j].numel() < need:
                pipe["ws"][j] = torch.empty(need, dtype=torch.uint8, device=dev)
        scan_cus = pipe["lanes"][0][2]
        o_scan = self._options(cu_limit=scan_cus, flags=_lib.TOPK_SCAN_ONLY)
        o_tail = self._options(cu_limit=tail_cus, flags=_lib.TOPK_TAIL_ONLY)
        xchg = pipe["xchg"]
        nb = len(batches)
        # every buffer the side streams touch is allocated HERE, on the caller's stream (nothing is allocated under, or recorded on, the
        # CU-masked streams: they are not torch's, and the caching allocator must never be asked to order a free against them)
        all_s = [torch.empty((q.shape[0], k), dtype=torch.float32, device=dev) for q in batches]
        all_i = [torch.empty((q.shape[0], k), dtype=torch.int64, device=dev) for q in batches]
        ready = torch.cuda.Event(); ready.record(cur)
        for scan, tail, _ in pipe["lanes"]:
            scan.wait_event(ready); tail.wait_event(ready)
        scanned = [torch.cuda.Event() for _ in range(nb)]
        tailed = [torch.cuda.Event() for _ in range(nb)]
        out = []
        do_dist = distributed and dist.is_initialized()
        if do_dist:
            xchg.wait_event(ready)
        for b, q in enumerate(batches):
            scan, tail, _ = pipe["lanes"][b % lanes]
            ws = pipe["ws"][b % lanes]
            if b >= lanes:
                scan.wait_event(tailed[b - lanes])           # the lane's workspace has been read out
            self.search(q, k, ws=ws, out=(all_s[b], all_i[b]), _opt=o_scan, _stream=scan.cuda_stream)
            scanned[b].record(scan)
            tail.wait_event(scanned[b])
            s, i = self.search(q, k, ws=ws, out=(all_s[b], all_i[b]), _opt=o_tail, _stream=tail.cuda_stream)
            tailed[b].record(tail)
            if do_dist:                                      # the exchange on an ordinary torch stream (RCCL's own kernels take any CU)
                xchg.wait_event(tailed[b])
                with torch.cuda.stream(xchg):
                    s, i = merge_partials(*gather_partials(s, i, group), k)
                    s.record_stream(cur); i.record_stream(cur)
            out.append((s, i))
        for scan, tail, _ in pipe["lanes"]:
            cur.wait_stream(scan); cur.wait_stream(tail)
        if do_dist:
            cur.wait_stream(xchg)
        self._last_ws = pipe["ws"][(nb - 1) % lanes]
        return out

    def search_distributed(self, queries_f16: torch.Tensor, k: int = 10, group=None):
        """Every rank passes the SAME queries; returns the global top-k on every rank."""
        import torch.distributed as dist
        s, i = self.search(queries_f16, k)
        if not dist.is_initialized():
            return s, i
        # world size 1 takes the same gather + merge path (a 1-part merge is the identity): one code path to test
        all_s, all_i = gather_partials(s, i, group)
        return merge_partials(all_s, all_i, k)


_CU_STREAMS = {}


def _cu_streams(device, tail_cus: int, lane: int = 0):
    """(scan stream, tail stream, CUs of the scan stream) for one device: two HIP streams with DISJOINT compute-unit masks
    (`arx_stream_create_cu_mask`) — `scan` on all but `tail_cus` CUs, `tail` on the rest; tail_cus a multiple of 8 (the driver deals mask
    bits round-robin to the 8 XCDs — measured: bits [224, 256) are 4 CUs on each XCD — so both streams keep CUs on every XCD).
    tail_cus = 0: two ordinary torch streams, the whole chip each.  Created once per (device, split) and kept for the life of the
    process: a HIP stream that torch has seen must not be destroyed while tensors or events may still refer to it."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(tail_cus), int(lane))
    if key in _CU_STREAMS:
        return _CU_STREAMS[key]
    lib = _lib.load()
    if tail_cus <= 0:
        val = (torch.cuda.Stream(device), torch.cuda.Stream(device), 0)
    else:
        with torch.cuda.device(device):
            n_cu = lib.arx_device_cu_count()
            assert tail_cus % 8 == 0 and tail_cus < n_cu, f"tail_cus={tail_cus} must be a multiple of 8 below {n_cu}"
            words = (n_cu + 31) // 32

            def stream_of(lo, hi):
                m = (C.c_uint32 * words)()
                for b in range(lo, hi):
                    m[b // 32] |= 1 << (b % 32)
                h = C.c_void_p(None)
                _lib.check(lib.arx_stream_create_cu_mask(m, words, C.byref(h)), "arx_stream_create_cu_mask")
                return torch.cuda.ExternalStream(h.value, device=device)
            val = (stream_of(0, n_cu - tail_cus), stream_of(n_cu - tail_cus, n_cu), n_cu - tail_cus)
    _CU_STREAMS[key] = val
    return val


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


def fill_clustered_rows(n_rows: int, dim: int, seed: int, n_clusters: int, device="cuda:0", row_base: int = 0, spread: float = 0.35,
                        hot_dims: int = 3, hot_gain: float = 6.0) -> torch.Tensor:
    """Embedding-LIKE synthetic rows generated in HBM (`arx_fill_clustered_rows_f16_at`): tight clusters around `n_clusters` centres and a
    few hot dimensions — what the iid Gaussian rows of `fill_unit_rows` are not: a query's neighbours score within the int8 bound's
    slack of each other, and max|x| (the int8 scale) is set by the hot dimensions.  Queries: same seed, a `row_base` beyond the corpus
    (scattered clusters) or inside it (topic order).  `n_clusters < 0`: TOPIC ORDER, consecutive runs of `-n_clusters` rows share a centre."""
    lib = _lib.load()
    t = torch.empty((n_rows, dim), dtype=torch.float16, device=device)
    if n_rows > 0:
        _lib.check(lib.arx_fill_clustered_rows_f16_at(t.data_ptr(), n_rows, dim, seed, row_base, n_clusters, spread, hot_dims, hot_gain,
                                                      torch.cuda.current_stream(t.device).cuda_stream), "arx_fill_clustered_rows_f16_at")
    return t
