"""Host side of the MI355X encoder: the drop-in for the object the reference calls `model`.

The reference's hot loop does (generate_embeddings_parallel.py:146-153)

    model.encode(batch, batch_size=..., normalize_embeddings=True, show_progress_bar=False,
                 convert_to_numpy=True, convert_to_tensor=False)

and `model.get_sentence_embedding_dimension()` (:169).  `HipSentenceEncoder` exposes exactly those
two methods with the same argument meaning, so that loop runs unchanged on top of it; underneath,
token batches go to the HIP kernels through the C ABI (include/arx.h).  PyTorch is used only for
device memory and streams.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .config import ARCH_MPNET, EncoderConfig
from .weights import layer_keys


def _dev_ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class HipEncoder:
    """Token ids -> unit-norm sentence embeddings on one GPU (one handle, one stream user)."""

    def __init__(self, cfg: EncoderConfig, state_dict: Dict[str, np.ndarray], device: Union[str, torch.device] = "cuda:0",
                 max_tokens: int = 32 * 512, max_seqs: int = 32):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.ArxError("no HIP device visible: the encoder has no CPU path")
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self._w: List[torch.Tensor] = []          # keeps device weights alive
        self._handle = C.c_void_p(None)
        self._cap = (0, 0)
        self._upload(state_dict)
        self._ensure_capacity(max_tokens, max_seqs)

    # ---- weights --------------------------------------------------------------------------------
    def _f32(self, a: np.ndarray) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        self._w.append(t)
        return t

    def _bf16(self, a: np.ndarray) -> torch.Tensor:
        # round-to-nearest-even on device via the library's own converter
        src = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=self.device)
        _lib.check(self.lib.arx_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(),
                                            torch.cuda.current_stream().cuda_stream), "arx_f32_to_bf16")
        self._w.append(dst)
        return dst

    def _upload(self, sd: Dict[str, np.ndarray]):
        cfg = self.cfg
        self._layers_c = (_lib.LayerWeightsC * cfg.layers)()
        for i in range(cfg.layers):
            k = layer_keys(cfg, i)
            wqkv = np.concatenate([sd[k["q"] + ".weight"], sd[k["k"] + ".weight"], sd[k["v"] + ".weight"]], 0)
            bqkv = np.concatenate([sd[k["q"] + ".bias"], sd[k["k"] + ".bias"], sd[k["v"] + ".bias"]], 0)
            L = self._layers_c[i]
            L.w_qkv = self._bf16(wqkv).data_ptr(); L.b_qkv = self._f32(bqkv).data_ptr()
            L.w_o = self._bf16(sd[k["o"] + ".weight"]).data_ptr(); L.b_o = self._f32(sd[k["o"] + ".bias"]).data_ptr()
            L.ln1_g = self._f32(sd[k["ln1"] + ".weight"]).data_ptr(); L.ln1_b = self._f32(sd[k["ln1"] + ".bias"]).data_ptr()
            L.w_fc1 = self._bf16(sd[k["fc1"] + ".weight"]).data_ptr(); L.b_fc1 = self._f32(sd[k["fc1"] + ".bias"]).data_ptr()
            L.w_fc2 = self._bf16(sd[k["fc2"] + ".weight"]).data_ptr(); L.b_fc2 = self._f32(sd[k["fc2"] + ".bias"]).data_ptr()
            L.ln2_g = self._f32(sd[k["ln2"] + ".weight"]).data_ptr(); L.ln2_b = self._f32(sd[k["ln2"] + ".bias"]).data_ptr()
        w = _lib.EncoderWeightsC()
        w.word_emb = self._f32(sd["embeddings.word_embeddings.weight"]).data_ptr()
        w.pos_emb = self._f32(sd["embeddings.position_embeddings.weight"]).data_ptr()
        w.emb_ln_g = self._f32(sd["embeddings.LayerNorm.weight"]).data_ptr()
        w.emb_ln_b = self._f32(sd["embeddings.LayerNorm.bias"]).data_ptr()
        if cfg.arch == ARCH_MPNET:
            w.type_emb = None
            w.rel_bias = self._f32(sd["encoder.relative_attention_bias.weight"]).data_ptr()
        else:
            w.type_emb = self._f32(sd["embeddings.token_type_embeddings.weight"][0]).data_ptr()
            w.rel_bias = None
        w.layers = C.cast(self._layers_c, C.POINTER(_lib.LayerWeightsC))
        self._weights_c = w
        c = _lib.EncoderConfigC(cfg.arch, cfg.vocab_size, cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.max_pos,
                                cfg.pool, cfg.pad_id, cfg.rel_buckets, cfg.rel_max_distance, cfg.ln_eps)
        self._cfg_c = c
        torch.cuda.synchronize(self.device)

    def _ensure_capacity(self, max_tokens: int, max_seqs: int):
        if max_tokens <= self._cap[0] and max_seqs <= self._cap[1]:
            return
        max_tokens = max(max_tokens, self._cap[0]); max_seqs = max(max_seqs, self._cap[1])
        self.close()
        h = C.c_void_p(None)
        _lib.check(self.lib.arx_encoder_create(C.byref(self._cfg_c), C.byref(self._weights_c), max_tokens, max_seqs,
                                               C.byref(h)), "arx_encoder_create")
        self._handle = h
        self._cap = (max_tokens, max_seqs)
        self._low_latency = False

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            torch.cuda.synchronize(self.device)
            self.lib.arx_encoder_destroy(self._handle)
            self._handle = C.c_void_p(None)
            self._cap = (0, 0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- forward --------------------------------------------------------------------------------
    def forward_tokens(self, ids: torch.Tensor, lens: torch.Tensor, max_len: int, total_tokens: int,
                       out: Optional[torch.Tensor] = None, out_f16: Optional[torch.Tensor] = None,
                       normalize: bool = True, low_latency: bool = False) -> Optional[torch.Tensor]:
        """ids int32 [B, S] (device), lens int32 [B] (device) -> f32 [B, H] (device).  Launches on the
        current torch stream; `out_f16` (fp16 [B, >=H], e.g. a slice of the corpus shard) is written in place.
        `low_latency`: a forward of <= 256 token rows (a query batch) runs the split-K small-batch schedule
        (`arx_encoder_set_low_latency`); rows then agree with the default schedule to rounding, not bit for bit."""
        B, S = ids.shape
        assert ids.dtype == torch.int32 and lens.dtype == torch.int32 and ids.is_contiguous() and lens.is_contiguous()
        self._ensure_capacity(total_tokens, B)
        if bool(low_latency) != self._low_latency:
            _lib.check(self.lib.arx_encoder_set_low_latency(self._handle, 1 if low_latency else 0), "arx_encoder_set_low_latency")
            self._low_latency = bool(low_latency)
        if out is None and out_f16 is None:
            out = torch.empty((B, self.cfg.hidden), dtype=torch.float32, device=self.device)
        rc = self.lib.arx_encoder_forward(
            self._handle, ids.data_ptr(), S, lens.data_ptr(), B, max_len, total_tokens,
            _dev_ptr(out), 0 if out is None else out.stride(0),
            _dev_ptr(out_f16), 0 if out_f16 is None else out_f16.stride(0),
            1 if normalize else 0, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "arx_encoder_forward")
        return out

    def encode_tokens(self, ids: np.ndarray, lens: np.ndarray, normalize: bool = True, low_latency: bool = False) -> torch.Tensor:
        """Right-padded ids [B, S] + lens [B] (host arrays) -> f32 [B, H] device tensor."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        if ids.shape[0] == 0:
            return torch.empty((0, self.cfg.hidden), dtype=torch.float32, device=self.device)
        max_len = max(int(lens.max()), 1)
        total = max(int(lens.sum()), 1)
        d_ids = torch.from_numpy(ids).to(self.device, non_blocking=True)
        d_lens = torch.from_numpy(lens).to(self.device, non_blocking=True)
        return self.forward_tokens(d_ids, d_lens, max_len, total, normalize=normalize, low_latency=low_latency)

    @staticmethod
    def _pad_batch(seqs: Sequence[Sequence[int]], idx: Sequence[int], pad_id: int):
        """Right-padded int32 [len(idx), S] + lens, built with one concatenate and one fancy assignment."""
        lens = np.fromiter((len(seqs[i]) for i in idx), np.int32, len(idx))
        S = max(int(lens.max()) if len(idx) else 1, 1)
        ids = np.full((len(idx), S), pad_id, np.int32)
        tot = int(lens.sum())
        if tot:
            flat = np.concatenate([np.asarray(seqs[i], np.int32) for i in idx if len(seqs[i])])
            rows = np.repeat(np.arange(len(idx)), lens)
            starts = np.cumsum(lens) - lens
            cols = np.arange(tot) - np.repeat(starts, lens)
            ids[rows, cols] = flat
        return ids, lens

    def encode_ragged(self, seqs: Sequence[Sequence[int]], batch_size: int = 256, normalize: bool = True,
                      on_device: bool = False, out_f16: Optional[torch.Tensor] = None, low_latency: bool = False):
        """Token-id lists -> f32 [n, H] numpy, input order preserved.  Sorted by length (descending, as
        sentence-transformers does) so each forward pads to a similar length; results do not depend on
        batch composition (key-padding mask), so the re-bucketing is invisible to the caller.  All forwards of the
        call are queued on the stream and the rows come back with ONE device->host copy (`on_device=True`: no copy,
        a device tensor in input order, for a consumer that also lives on the GPU).  `out_f16` (device fp16 [n, >= H], e.g. a
        slice of the rank's corpus shard) additionally receives the kernel's own fp16 rows, in input order: the shard is filled
        where it will be searched and never crosses PCIe."""
        n = len(seqs)
        if n == 0:
            return (torch.zeros((0, self.cfg.hidden), dtype=torch.float32, device=self.device) if on_device
                    else np.zeros((0, self.cfg.hidden), np.float32))
        order = sorted(range(n), key=lambda i: -len(seqs[i]))
        dev_out = torch.empty((n, self.cfg.hidden), dtype=torch.float32, device=self.device)
        dev16 = self._f16_stage(n, out_f16)
        for s0 in range(0, n, batch_size):
            idx = order[s0:s0 + batch_size]
            ids, lens = self._pad_batch(seqs, idx, self.cfg.pad_id)
            d_ids = torch.from_numpy(ids).to(self.device, non_blocking=True)
            d_lens = torch.from_numpy(lens).to(self.device, non_blocking=True)
            self.forward_tokens(d_ids, d_lens, ids.shape[1], max(int(lens.sum()), 1), out=dev_out[s0:s0 + len(idx)],
                                out_f16=None if dev16 is None else dev16[s0:s0 + len(idx)], normalize=normalize,
                                low_latency=low_latency)
        order_t = torch.as_tensor(order, device=self.device) if (on_device or dev16 is not None) else None
        if dev16 is not None:
            out_f16[:, :self.cfg.hidden].index_copy_(0, order_t, dev16)
        if on_device:
            out = torch.empty_like(dev_out)
            out[order_t] = dev_out
            return out
        out = np.empty((n, self.cfg.hidden), np.float32)
        out[np.asarray(order)] = dev_out.cpu().numpy()
        return out

    def _f16_stage(self, n: int, out_f16: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """Length-sorted staging rows for the fp16 output (the forwards write sorted order; one index_copy un-sorts)."""
        if out_f16 is None:
            return None
        assert out_f16.is_cuda and out_f16.dtype == torch.float16 and out_f16.dim() == 2 and out_f16.stride(1) == 1
        assert out_f16.shape[0] == n and out_f16.shape[1] >= self.cfg.hidden, "out_f16 must be [n, >= hidden]"
        return torch.empty((n, self.cfg.hidden), dtype=torch.float16, device=self.device)

    def encode_packed(self, ids: np.ndarray, lens: np.ndarray, batch_size: int = 256, normalize: bool = True,
                      on_device: bool = False, out_f16: Optional[torch.Tensor] = None, low_latency: bool = False):
        """`encode_ragged` for a tokenizer that already produced a right-padded id matrix (int32 [n, W]) and lengths: no
        per-token Python objects anywhere on the host path.  Same length-sorted batching, same rows."""
        n = int(len(lens))
        if n == 0:
            return (torch.zeros((0, self.cfg.hidden), dtype=torch.float32, device=self.device) if on_device
                    else np.zeros((0, self.cfg.hidden), np.float32))
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        order = np.argsort(-lens.astype(np.int64), kind="stable")
        dev_out = torch.empty((n, self.cfg.hidden), dtype=torch.float32, device=self.device)
        dev16 = self._f16_stage(n, out_f16)
        for s0 in range(0, n, batch_size):
            idx = order[s0:s0 + batch_size]
            bl = lens[idx]
            ml = max(int(bl.max()), 1)
            d_ids = torch.from_numpy(np.ascontiguousarray(ids[idx, :ml], dtype=np.int32)).to(self.device, non_blocking=True)
            d_lens = torch.from_numpy(np.ascontiguousarray(bl)).to(self.device, non_blocking=True)
            self.forward_tokens(d_ids, d_lens, ml, max(int(bl.sum()), 1), out=dev_out[s0:s0 + len(idx)],
                                out_f16=None if dev16 is None else dev16[s0:s0 + len(idx)], normalize=normalize,
                                low_latency=low_latency)
        order_t = torch.from_numpy(order).to(self.device) if (on_device or dev16 is not None) else None
        if dev16 is not None:
            out_f16[:, :self.cfg.hidden].index_copy_(0, order_t, dev16)
        if on_device:
            out = torch.empty_like(dev_out)
            out[order_t] = dev_out
            return out
        out = np.empty((n, self.cfg.hidden), np.float32)
        out[order] = dev_out.cpu().numpy()
        return out

    def tap_hidden(self, ids: np.ndarray, lens: np.ndarray, layer: int) -> np.ndarray:
        """Parity tap: packed hidden state [sum(lens), H] f32 after `layer` (0 = embeddings)."""
        _lib.check(self.lib.arx_encoder_set_tap(self._handle, layer), "arx_encoder_set_tap")
        self.encode_tokens(ids, lens)
        T = int(np.asarray(lens).sum())
        dst = torch.empty((T, self.cfg.hidden), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.arx_encoder_debug_hidden(self._handle, layer, dst.data_ptr(), T,
                                                     torch.cuda.current_stream().cuda_stream), "arx_encoder_debug_hidden")
        _lib.check(self.lib.arx_encoder_set_tap(self._handle, -1), "arx_encoder_set_tap")
        return dst.cpu().numpy()


def adjacent_cosines(embeddings: torch.Tensor) -> torch.Tensor:
    """cos(e_i, e_{i+1}) for consecutive rows (device f32 [n, D] -> [n-1]) — the similarity the stage-3 semantic
    chunker breaks on when it drops below 0.7 (text_processor.py:1555-1561)."""
    lib = _lib.load()
    e = embeddings
    assert e.is_cuda and e.dtype == torch.float32 and e.dim() == 2 and e.stride(1) == 1
    n, d = e.shape
    out = torch.empty((max(n - 1, 0),), dtype=torch.float32, device=e.device)
    _lib.check(lib.arx_adjacent_cosine(e.data_ptr(), e.stride(0), n, d, out.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "arx_adjacent_cosine")
    return out


class HipSentenceEncoder:
    """`SentenceTransformer`-shaped facade: tokenizer + HipEncoder.

    encode(sentences, batch_size=32, normalize_embeddings=False, convert_to_numpy=True,
           convert_to_tensor=False, show_progress_bar=None)  -> np.ndarray [n, D] float32
    mirrors the call at generate_embeddings_parallel.py:146-153 / :160-165."""

    def __init__(self, cfg: EncoderConfig, state_dict, tokenizer, device="cuda:0", max_batch: int = 1024):
        self.cfg = cfg
        self.tokenizer = tokenizer
        self.max_seq_length = cfg.max_seq_length
        self.encoder = HipEncoder(cfg, state_dict, device=device)
        self.max_batch = max_batch
        self.slab_texts = 8192          # texts tokenised per feeder step (steady state)
        self.first_slab_texts = 1024    # the first step is small so the GPU starts early; steps double up to slab_texts
        # `batch_size` only bounds memory in sentence-transformers; rows do not depend on batch composition here (key-padding mask,
        # token-packed activations; asserted by the parity tests), so small caller batches are coalesced into max_batch-sequence
        # forwards: better tile quantisation on 256 CUs and 5x fewer launches at the reference's default of 200
        self.coalesce_batches = True

    def get_sentence_embedding_dimension(self) -> int:
        return self.cfg.hidden

    def tokenize(self, sentences: Sequence[str]) -> List[List[int]]:
        return self.tokenizer.encode_batch(list(sentences), self.max_seq_length)

    def _tokenize_any(self, sentences: Sequence[str]):
        """-> ("packed", ids, lens) from a tokenizer with `encode_batch_packed` (the native feeder), else ("ragged", lists)."""
        if hasattr(self.tokenizer, "encode_batch_packed"):
            ids, lens = self.tokenizer.encode_batch_packed(list(sentences), self.max_seq_length)
            return ("packed", ids, lens)
        return ("ragged", self.tokenize(sentences))

    def _encode_tokens_any(self, toks, bs: int, normalize: bool, on_device: bool = False, out_f16=None, low_latency: bool = False):
        if toks[0] == "packed":
            return self.encoder.encode_packed(toks[1], toks[2], batch_size=bs, normalize=normalize, on_device=on_device,
                                              out_f16=out_f16, low_latency=low_latency)
        return self.encoder.encode_ragged(toks[1], batch_size=bs, normalize=normalize, on_device=on_device, out_f16=out_f16,
                                          low_latency=low_latency)

    def encode_device(self, sentences: Sequence[str], batch_size: int = 32, normalize_embeddings: bool = False) -> torch.Tensor:
        """Rows stay in HBM (f32 [n, D], input order) — for GPU-side consumers (adjacent cosine, the search index)."""
        bs = max(1, min(batch_size, self.max_batch))
        return self._encode_tokens_any(self._tokenize_any(list(sentences)), bs, normalize_embeddings, on_device=True)

    def encode(self, sentences, batch_size: int = 32, show_progress_bar=None, convert_to_numpy: bool = True,
               convert_to_tensor: bool = False, normalize_embeddings: bool = False, device_f16_out: Optional[torch.Tensor] = None,
               low_latency: bool = False, **_ignored):
        """`device_f16_out` (additive; device fp16 [n, >= D]): the same rows, as the kernel's own fp16 output, are also left in
        HBM in input order — the CLI passes a slice of the rank's corpus shard, so the search step never re-uploads them.
        `low_latency` (additive): for QUERY texts — a forward of <= 256 token rows takes the small-batch schedule
        (`arx_encoder_set_low_latency`: 1.5 -> 0.49 ms for one query at the mpnet-base shape); rows agree with the default to rounding."""
        single = isinstance(sentences, str)
        if single:
            sentences = [sentences]
        sentences = list(sentences)
        bs = self.max_batch if self.coalesce_batches else max(1, min(batch_size, self.max_batch))
        slab = max(bs, self.slab_texts)
        first = max(bs, self.first_slab_texts)
        if len(sentences) <= first:
            emb = self._encode_tokens_any(self._tokenize_any(sentences), bs, normalize_embeddings, out_f16=device_f16_out,
                                          low_latency=low_latency)
        else:
            # feeder: the tokenizer (native C++ or Rust; both release the GIL) works on slab i+1 in a helper thread while the GPU encodes slab i;
            # slabs grow 1024, 2048, ... up to slab_texts so the GPU starts after ~25 ms of tokenisation, not a whole slab
            from concurrent.futures import ThreadPoolExecutor
            bounds, size, s0 = [], first, 0
            while s0 < len(sentences):
                bounds.append((s0, min(len(sentences), s0 + size)))
                s0 += size
                size = min(slab, size * 2)
            parts = []
            with ThreadPoolExecutor(max_workers=1) as ex:
                fut = ex.submit(self._tokenize_any, sentences[bounds[0][0]:bounds[0][1]])
                for bi in range(len(bounds)):
                    seqs = fut.result()
                    if bi + 1 < len(bounds):
                        fut = ex.submit(self._tokenize_any, sentences[bounds[bi + 1][0]:bounds[bi + 1][1]])
                    o16 = None if device_f16_out is None else device_f16_out[bounds[bi][0]:bounds[bi][1]]
                    parts.append(self._encode_tokens_any(seqs, bs, normalize_embeddings, out_f16=o16))
            emb = np.concatenate(parts, 0)
        if convert_to_tensor:
            emb = torch.from_numpy(emb)
        return emb[0] if single else emb
