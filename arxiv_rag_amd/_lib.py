"""ctypes binding of libarx_hip.so (C ABI: include/arx.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load this
raises, and every caller above it fails with it.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("ARX_LIB", _HERE / "libarx_hip.so"))

K_CLASSES = ["search_groupmax", "gemm_qkv", "gemm_oproj", "gemm_fc1", "gemm_fc2", "attention",
             "layernorm", "embed", "pool", "search_select", "search_rescore", "gemm_raw"]


class EncoderConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("arch", "vocab_size", "hidden", "layers", "heads", "ffn", "max_pos",
                                         "pool", "pad_id", "rel_buckets", "rel_max_distance")] + [("ln_eps", C.c_float)]


class LayerWeightsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_qkv", "b_qkv", "w_o", "b_o", "ln1_g", "ln1_b",
                                          "w_fc1", "b_fc1", "w_fc2", "b_fc2", "ln2_g", "ln2_b")]


class EncoderWeightsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b", "rel_bias")] \
               + [("layers", C.POINTER(LayerWeightsC))]


class TopkOptionsC(C.Structure):
    """arx_topk_options (include/arx.h): the per-call search policy; the library keeps none of its own."""
    _fields_ = [("struct_bytes", C.c_int32), ("i8_max_queries", C.c_int32), ("max_row_norm", C.c_float), ("cu_limit", C.c_int32),
                ("flags", C.c_int32), ("debug_tau_mult", C.c_float), ("debug_drop_best", C.c_int32)]

    def __init__(self, **kw):
        super().__init__(**kw)
        self.struct_bytes = C.sizeof(TopkOptionsC)


TOPK_NO_PERSISTENT, TOPK_SCAN_ONLY, TOPK_TAIL_ONLY, TOPK_NO_SINGLE_ROW_TAIL, TOPK_I8_CENTRE_QUERY = 1, 2, 4, 8, 16


EXPORTS = {
    # name: (restype, argtypes)
    "arx_version": (C.c_int32, []),
    "arx_last_error": (C.c_char_p, []),
    "arx_build_info": (C.c_int32, []),
    "arx_mpnet_bucket": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "arx_encoder_workspace_bytes": (C.c_int64, [C.POINTER(EncoderConfigC), C.c_int32, C.c_int32]),
    "arx_encoder_create": (C.c_int32, [C.POINTER(EncoderConfigC), C.POINTER(EncoderWeightsC), C.c_int32, C.c_int32,
                                       C.POINTER(C.c_void_p)]),
    "arx_encoder_destroy": (None, [C.c_void_p]),
    "arx_encoder_forward": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "arx_encoder_attention": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "arx_encoder_set_tap": (C.c_int32, [C.c_void_p, C.c_int32]),
    "arx_encoder_set_low_latency": (C.c_int32, [C.c_void_p, C.c_int32]),
    "arx_encoder_debug_hidden": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "arx_topk_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "arx_topk_search": (C.c_int32, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "arx_topk_i8_index_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "arx_topk_i8_index_info": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_float), C.c_void_p]),
    "arx_topk_build_i8": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "arx_topk_search_i8": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "arx_topk_workspace_bytes_i8": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "arx_topk_search_opt": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(TopkOptionsC), C.c_void_p]),
    "arx_rows_max_norm_f16": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "arx_stream_create_cu_mask": (C.c_int32, [C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_void_p)]),
    "arx_stream_destroy": (C.c_int32, [C.c_void_p]),
    "arx_device_cu_count": (C.c_int32, []),
    "arx_debug_cu_census": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "arx_topk_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "arx_topk_merge": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "arx_gemm_bf16": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p]),
    "arx_prof_classes": (C.c_int32, [C.c_uint32]),
    "arx_wp_create": (C.c_int32, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_char_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "arx_wp_destroy": (None, [C.c_void_p]),
    "arx_wp_encode": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "arx_wp_miss_count": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "arx_wp_miss_fetch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "arx_wp_cache_add": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "arx_wp_cache_size": (C.c_int64, [C.c_void_p]),
    "arx_wp_version": (C.c_int32, []),
    "arx_adjacent_cosine": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "arx_f32_to_bf16": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "arx_fill_unit_rows_f16": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_void_p]),
    "arx_fill_unit_rows_f16_at": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_int64, C.c_void_p]),
    "arx_fill_clustered_rows_f16_at": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_int64, C.c_int32, C.c_float, C.c_int32,
                                                   C.c_float, C.c_void_p]),
    "arx_prof_enable": (C.c_int32, [C.c_int32]),
    "arx_prof_reset": (C.c_int32, []),
    "arx_prof_read": (C.c_int32, [C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
}

_lib = None


class ArxError(RuntimeError):
    pass


def load():
    """Load libarx_hip.so once; raise (never fall back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ArxError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or arxiv_rag_amd/csrc/build.sh). There is no CPU fallback for this path.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().arx_last_error().decode("utf-8", "replace")
        raise ArxError(f"{what} failed (rc={rc}): {msg}")


def prof_classes(names=None):
    """Restrict event recording to these kernel classes (None = all)."""
    mask = 0xFFFFFFFF if names is None else sum(1 << K_CLASSES.index(n) for n in names)
    check(load().arx_prof_classes(mask), "arx_prof_classes")


def prof_enable(on: bool):
    check(load().arx_prof_enable(1 if on else 0), "arx_prof_enable")


def prof_reset():
    check(load().arx_prof_reset(), "arx_prof_reset")


def prof_read():
    """{kernel class: (total_ms, launches)} — synchronises on the recorded events."""
    lib = load()
    out = {}
    for i, name in enumerate(K_CLASSES):
        ms, n = C.c_float(0), C.c_int32(0)
        check(lib.arx_prof_read(i, C.byref(ms), C.byref(n)), "arx_prof_read")
        out[name] = (float(ms.value), int(n.value))
    return out
