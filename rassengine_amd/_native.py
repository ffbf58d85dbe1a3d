"""ctypes binding of ``librass_hip.so`` (the C ABI declared in ``include/rass_engine.h``).

There is NO fallback: if the HIP library is missing or fails to load, importing the
product path raises.  A CPU stand-in exists only under ``oracle/`` for the tests.

H1 of SURVEY §7 (two HIP runtimes): ``torch 2.10+rocm7.0`` ships its own
``libamdhip64.so`` with the same soname (``libamdhip64.so.7``) as ROCm 7.2's.  We import
torch first so the HIP runtime torch uses is the one already mapped when
``librass_hip.so`` resolves its ``DT_NEEDED``; device pointers from torch tensors and from
the engine then belong to one runtime.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# RASS_HIP_LIB: A/B another build of the same ABI (kernel experiments); the default is the in-tree library
LIB_PATH = os.environ.get("RASS_HIP_LIB") or os.path.join(_HERE, "lib", "librass_hip.so")

RASS_OK = 0
RASS_MAX_K = 32
RASS_MAX_K_MULTIPASS = 4096
RASS_TAG_PATIENT_MASK = 0x00FFFFFF
RASS_TAG_DOCTYPE_SHIFT = 24
RASS_TAG_DOCTYPE_MASK = 0x7F000000
RASS_MAX_QBATCH = 32
RASS_F32 = 0
RASS_BF16 = 1
RASS_QFILTER_NONE = -1

c_float_p = ctypes.POINTER(ctypes.c_float)
c_i32_p = ctypes.POINTER(ctypes.c_int32)
c_i64_p = ctypes.POINTER(ctypes.c_int64)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); the export test walks this table against the header.
SIGNATURES = {
    "rass_abi_version": (ctypes.c_int, []),
    "rass_last_error": (ctypes.c_char_p, []),
    "rass_device_count": (ctypes.c_int, []),
    "rass_engine_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_void_pp]),
    "rass_engine_destroy": (None, [ctypes.c_void_p]),
    "rass_engine_dim": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_engine_device": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_engine_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rass_engine_reset_stream": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_engine_get_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "rass_engine_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_index_open": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int64, c_void_pp]),
    "rass_index_drop": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "rass_index_count": (ctypes.c_int64, [ctypes.c_void_p]),
    "rass_index_rows": (ctypes.c_int64, [ctypes.c_void_p]),
    "rass_index_dtype": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_index_has_global_ids": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_index_dim": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_index_row_stride": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_index_add": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                      ctypes.c_int, c_i64_p]),
    "rass_index_add_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.c_int, c_i64_p]),
    "rass_index_add_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_i64_p]),
    "rass_index_delete": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    "rass_index_get_row": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rass_index_get_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "rass_index_search": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rass_index_search_multi": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rass_index_search_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rass_index_search_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "rass_index_search_device_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                   ctypes.c_void_p]),
    "rass_index_search_device_after": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_void_p]),
    "rass_index_search_device_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_int64, ctypes.c_int64]),
    "rass_index_set_prefilter": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rass_index_get_prefilter": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_index_candidates_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_void_p]),
    "rass_index_save": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "rass_index_load": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, c_void_pp]),
    "rass_index_fill_synthetic": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64,
                                                 ctypes.c_int64]),
    "rass_scan_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "rass_scan_topk_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "rass_pack_rows_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p]),
    "rass_unpack_rows_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rass_gather_rows_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "rass_topk_merge": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rass_topk_merge_strided": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p]),
    "rass_topk_merge_strided_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                     ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                     ctypes.c_void_p]),
    "rass_normalize_rows_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "rass_peer_buffer_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_size_t, c_void_pp, ctypes.c_char_p]),
    "rass_peer_buffer_open": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, c_void_pp]),
    "rass_peer_buffer_close": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rass_peer_post": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_uint64, ctypes.c_void_p]),
    "rass_peer_wait": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_void_p]),
    "rass_kmeans_assign": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "rass_kmeans_accumulate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "rass_ivf_save": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "rass_ivf_load": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_void_pp]),
    "rass_ivf_build": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, c_void_pp]),
    "rass_ivf_destroy": (None, [ctypes.c_void_p]),
    "rass_ivf_build_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                         c_void_pp]),
    "rass_ivf_dtype": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_ivf_rows": (ctypes.c_int64, [ctypes.c_void_p]),
    "rass_ivf_nlist": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_ivf_search": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_i64_p]),
    "rass_ivf_search_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rass_ivf_search_device_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p]),
    "rass_ivf_build_prefix": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int64, c_void_pp]),
    "rass_ivf_covered_rows": (ctypes.c_int64, [ctypes.c_void_p]),
    "rass_ivf_delete": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    "rass_ivf_search_delta": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, c_i64_p]),
    "rass_ivf_search_delta_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_void_p]),
    "rass_encoder_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, c_void_pp]),
    "rass_encoder_destroy": (None, [ctypes.c_void_p]),
    "rass_encoder_hidden": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_encoder_get_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "rass_encoder_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rass_encoder_set_weight": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64]),
    "rass_encoder_finalize": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_encode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "rass_encode_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "rass_gemm_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_void_p]),
    "rass_gemm_bf16_ws": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "rass_attention_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "rass_attention_out_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_void_p]),
    "rass_tokenizer_create": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, c_void_pp]),
    "rass_tokenizer_destroy": (None, [ctypes.c_void_p]),
    "rass_tokenizer_vocab_size": (ctypes.c_int, [ctypes.c_void_p]),
    "rass_tokenizer_encode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int,
                                             ctypes.c_void_p]),
    "rass_tokenizer_encode_batch": (ctypes.c_int64, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p),
                                                     ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                     ctypes.c_void_p, ctypes.c_int]),
    "rass_timer_create": (ctypes.c_int, [c_void_pp]),
    "rass_timer_destroy": (None, [ctypes.c_void_p]),
    "rass_timer_start": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rass_timer_stop": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rass_timer_elapsed_ms": (ctypes.c_int, [ctypes.c_void_p, c_float_p]),
    "rass_engine_kernel_timing_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rass_engine_kernel_timing_end": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                                     ctypes.POINTER(ctypes.c_int)]),
    "rass_index_device_rows": (ctypes.c_void_p, [ctypes.c_void_p]),
    "rass_index_device_tags": (ctypes.c_void_p, [ctypes.c_void_p]),
    "rass_scan_kernel_name": (ctypes.c_char_p, [ctypes.c_int, ctypes.c_int]),
}

_lib: Optional[ctypes.CDLL] = None


class RassError(RuntimeError):
    """A C-ABI call returned a negative rass_status."""

    def __init__(self, fn: str, code: int, text: str):
        super().__init__(f"{fn} failed ({code}): {text}")
        self.code = code


def lib() -> ctypes.CDLL:
    """Load librass_hip.so once.  Raises (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C rassengine_amd/csrc`. rassengine_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  -- map torch's HIP runtime first (see module docstring)
    except Exception:  # torch-free C/ctypes users link against the system ROCm instead
        pass
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError = header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    if L.rass_abi_version() != 1:
        raise ImportError(f"librass_hip.so ABI {L.rass_abi_version()} != 1")
    _lib = L
    return L


def check(fn: str, code: int) -> int:
    if code < 0:
        raise RassError(fn, code, lib().rass_last_error().decode("utf-8", "replace"))
    return code
