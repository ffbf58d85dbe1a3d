"""Stateless launchers on torch device tensors (pointer carriers): the fused scan+top-k,
the top-k merge and the reference normalise.  Everything runs on torch's current stream so
that ``torch.cuda.Event`` / ``torch.distributed`` ordering just works.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _native as N

_workspaces = {}


def _stream_ptr() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _ws(device: torch.device, nbytes: int) -> torch.Tensor:
    key = (device.index, _stream_ptr())
    w = _workspaces.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = w
    return w


def _req(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must live in HBM (a cuda tensor); rassengine_amd has no CPU path")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def normalize_rows(x: torch.Tensor, out_stride: Optional[int] = None) -> torch.Tensor:
    """out = x / (||x|| + 1e-9) row-wise (reference app/main.py:1249-1251), zero padded to
    ``out_stride`` columns."""
    _req(x, torch.float32, "x")
    n, d = x.shape
    stride = d if out_stride is None else int(out_stride)
    out = torch.empty((n, stride), dtype=torch.float32, device=x.device)
    N.check("rass_normalize_rows_f32",
            N.lib().rass_normalize_rows_f32(ctypes.c_void_p(x.data_ptr()), d, ctypes.c_void_p(out.data_ptr()),
                                            stride, n, d, ctypes.c_void_p(_stream_ptr())))
    return out


def pack_rows(x: torch.Tensor, normalize: bool = False, row_stride: Optional[int] = None) -> torch.Tensor:
    """Row-major fp32 [n, dim] -> tile16 slab [ceil(n/16)*16, row_stride] (the HBM layout the
    scan streams; see include/rass_engine.h).  ``normalize`` applies the reference formula."""
    _req(x, torch.float32, "x")
    n, d = x.shape
    # whole 128-column units up to 1 024 columns, whole 256-column units above (the wide-row scan's panels)
    stride = ((d + 127) // 128 * 128 if d <= 1024 else (d + 255) // 256 * 256) if row_stride is None else int(row_stride)
    packed = torch.zeros(((n + 15) // 16 * 16, stride), dtype=torch.float32, device=x.device)
    N.check("rass_pack_rows_f32",
            N.lib().rass_pack_rows_f32(ctypes.c_void_p(x.data_ptr()), d, ctypes.c_void_p(packed.data_ptr()), stride,
                                       0, n, d, 1 if normalize else 0, ctypes.c_void_p(_stream_ptr())))
    return packed


def unpack_rows(packed: torch.Tensor, n: int, dim: int, first_row: int = 0) -> torch.Tensor:
    """tile16 slab -> row-major fp32 [n, dim] (rows first_row .. first_row+n)."""
    _req(packed, torch.float32, "packed")
    out = torch.empty((n, dim), dtype=torch.float32, device=packed.device)
    N.check("rass_unpack_rows_f32",
            N.lib().rass_unpack_rows_f32(ctypes.c_void_p(packed.data_ptr()), packed.shape[1], int(first_row), n, dim,
                                         ctypes.c_void_p(out.data_ptr()), dim, ctypes.c_void_p(_stream_ptr())))
    return out


def scan_topk_packed(packed: torch.Tensor, n_rows: int, queries: torch.Tensor, k: int, *,
                     row_tag: Optional[torch.Tensor] = None, q_filter: Optional[torch.Tensor] = None,
                     id_base: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact cosine top-k of ``queries`` [nq<=32, dim] over a normalised tile16 slab holding
    ``n_rows`` rows.  Returns (scores f32 [nq,k], ids i64 [nq,k]) on the device, async on the
    current stream."""
    _req(packed, torch.float32, "packed")
    _req(queries, torch.float32, "queries")
    stride = packed.shape[1]
    nq, d = queries.shape
    if packed.shape[0] % 16 != 0 or packed.shape[0] < n_rows:
        raise ValueError("packed slab must hold whole 16-row blocks covering n_rows")
    if row_tag is not None:
        _req(row_tag, torch.int32, "row_tag")
    if q_filter is not None:
        _req(q_filter, torch.int32, "q_filter")
    L = N.lib()
    ws_bytes = int(L.rass_scan_workspace_bytes(nq, k))
    if ws_bytes == 0:
        raise ValueError(f"unsupported (nq={nq}, k={k}); nq<=32, k<=32")
    ws = _ws(packed.device, ws_bytes)
    out_s = torch.empty((nq, k), dtype=torch.float32, device=packed.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=packed.device)
    N.check("rass_scan_topk_f32",
            L.rass_scan_topk_f32(ctypes.c_void_p(packed.data_ptr()), int(n_rows), d, stride,
                                 ctypes.c_void_p(row_tag.data_ptr() if row_tag is not None else 0),
                                 ctypes.c_void_p(queries.data_ptr()), nq,
                                 ctypes.c_void_p(q_filter.data_ptr() if q_filter is not None else 0), k,
                                 int(id_base), ctypes.c_void_p(out_s.data_ptr()), ctypes.c_void_p(out_i.data_ptr()),
                                 ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(_stream_ptr())))
    return out_s, out_i


def scan_topk(corpus: torch.Tensor, queries: torch.Tensor, k: int, *, row_tag: Optional[torch.Tensor] = None,
              q_filter: Optional[torch.Tensor] = None, id_base: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Convenience for tests: ``corpus`` is ROW-MAJOR normalised fp32 [n, dim]; it is packed
    to tile16 first (an index keeps its rows packed, so the product path never does this)."""
    n = corpus.shape[0]
    packed = pack_rows(corpus, normalize=False)
    return scan_topk_packed(packed, n, queries, k, row_tag=row_tag, q_filter=q_filter, id_base=id_base)


def topk_merge(list_scores: torch.Tensor, list_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge [n_lists, nq, k] candidate lists -> [nq, k] under (score desc, id asc)."""
    _req(list_scores, torch.float32, "list_scores")
    _req(list_ids, torch.int64, "list_ids")
    n_lists, nq, k = list_scores.shape
    out_s = torch.empty((nq, k), dtype=torch.float32, device=list_scores.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=list_scores.device)
    N.check("rass_topk_merge",
            N.lib().rass_topk_merge(ctypes.c_void_p(list_scores.data_ptr()), ctypes.c_void_p(list_ids.data_ptr()),
                                    n_lists, nq, k, ctypes.c_void_p(out_s.data_ptr()),
                                    ctypes.c_void_p(out_i.data_ptr()), ctypes.c_void_p(_stream_ptr())))
    return out_s, out_i


# ------------------------------------------------------------------------------------------- the encoder's kernels
def gemm_bf16(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, residual: Optional[torch.Tensor] = None,
              epilogue: int = 0) -> torch.Tensor:
    """``rass_gemm_bf16``: y = epi(x w^T + bias), bf16 operands, fp32 accumulation; epilogue 0 bias, 1 bias + residual,
    2 bias + GELU(erf).  x [m, k], w [n, k] (the encoder's weight layout), bias fp32 [n]; n % 128 == 0, k % 64 == 0."""
    _req(x, torch.bfloat16, "x")
    _req(w, torch.bfloat16, "w")
    _req(bias, torch.float32, "bias")
    m, k = x.shape
    n = w.shape[0]
    if w.shape[1] != k or bias.shape != (n,) or n % 128 or k % 64:
        raise ValueError(f"gemm_bf16: x [m, k], w [n, k], bias [n] with n % 128 == 0 and k % 64 == 0; got {tuple(x.shape)}, {tuple(w.shape)}")
    if epilogue not in (0, 1, 2) or (epilogue == 1) != (residual is not None):
        raise ValueError("gemm_bf16: epilogue 1 (and only it) takes a residual")
    m_pad = (m + 255) // 256 * 256 if m >= 1024 else (m + 127) // 128 * 128     # whole tiles must be allocated
    xin = x
    if m_pad != m:
        xin = torch.zeros((m_pad, k), dtype=torch.bfloat16, device=x.device)
        xin[:m] = x
    r = None
    if residual is not None:
        _req(residual, torch.bfloat16, "residual")
        if residual.shape != (m, n):
            raise ValueError("gemm_bf16: residual must be [m, n]")
        r = residual
        if m_pad != m:
            r = torch.zeros((m_pad, n), dtype=torch.bfloat16, device=x.device)
            r[:m] = residual
    y = torch.empty((m_pad, n), dtype=torch.bfloat16, device=x.device)
    N.check("rass_gemm_bf16",
            N.lib().rass_gemm_bf16(ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
                                   ctypes.c_void_p(r.data_ptr()) if r is not None else None, ctypes.c_void_p(y.data_ptr()),
                                   int(m), int(m_pad), int(n), int(k), int(epilogue), ctypes.c_void_p(_stream_ptr())))
    return y[:m]


def attention_bf16(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int, heads: int) -> torch.Tensor:
    """``rass_attention_bf16``: varlen multi-head self-attention, heads of 64, softmax(q k^T / 8) v in fp32.  qkv bf16
    [tokens, 3 * hidden] (q | k | v per token), cu_seqlens int32 [nseq + 1] (device) -> ctx bf16 [tokens, hidden]."""
    _req(qkv, torch.bfloat16, "qkv")
    _req(cu_seqlens, torch.int32, "cu_seqlens")
    tokens, three_h = qkv.shape
    hidden = three_h // 3
    if three_h != 3 * hidden or hidden != 64 * int(heads) or not 1 <= int(max_seqlen) <= 512:
        raise ValueError("attention_bf16: qkv [tokens, 3 * 64 * heads], max_seqlen <= 512")
    nseq = cu_seqlens.numel() - 1
    ctx = torch.empty((tokens, hidden), dtype=torch.bfloat16, device=qkv.device)
    N.check("rass_attention_bf16",
            N.lib().rass_attention_bf16(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(cu_seqlens.data_ptr()), int(nseq),
                                        int(tokens), int(max_seqlen), int(hidden), int(heads), ctypes.c_void_p(ctx.data_ptr()),
                                        ctypes.c_void_p(_stream_ptr())))
    return ctx


def attention_out_bf16(qkv: torch.Tensor, cu_seqlens: torch.Tensor, heads: int, w: torch.Tensor, bias: torch.Tensor,
                       residual: torch.Tensor) -> torch.Tensor:
    """``rass_attention_out_bf16``: query time — attention and its output projection in ONE launch,
    ``attention(qkv) @ w.T + bias + residual`` (1..32 tokens over all sequences, hidden 1024, 16 heads; anything else
    raises: the library answers RASS_ERR_UNSUPPORTED).  qkv bf16 [tokens, 3 * hidden], w bf16 [n, hidden], bias fp32 [n],
    residual bf16 [tokens, n] -> bf16 [tokens, n]."""
    _req(qkv, torch.bfloat16, "qkv")
    _req(cu_seqlens, torch.int32, "cu_seqlens")
    _req(w, torch.bfloat16, "w")
    _req(bias, torch.float32, "bias")
    _req(residual, torch.bfloat16, "residual")
    tokens, three_h = qkv.shape
    hidden = three_h // 3
    n = w.shape[0]
    if three_h != 3 * hidden or w.shape[1] != hidden or bias.shape[0] != n or tuple(residual.shape) != (tokens, n):
        raise ValueError("attention_out_bf16: qkv [tokens, 3 hidden], w [n, hidden], bias [n], residual [tokens, n]")
    nseq = int(cu_seqlens.shape[0]) - 1
    y = torch.empty((tokens, n), dtype=torch.bfloat16, device=qkv.device)
    N.check("rass_attention_out_bf16",
            N.lib().rass_attention_out_bf16(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(cu_seqlens.data_ptr()), nseq,
                                            int(tokens), int(hidden), int(heads), ctypes.c_void_p(w.data_ptr()),
                                            ctypes.c_void_p(bias.data_ptr()), ctypes.c_void_p(residual.data_ptr()),
                                            ctypes.c_void_p(y.data_ptr()), int(n), ctypes.c_void_p(_stream_ptr())))
    return y


def encode(encoder_handle: int, token_ids: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int, hidden: int
           ) -> torch.Tensor:
    """``rass_encode_device`` on torch's current stream: the whole sentence-encoder forward (embeddings + LayerNorm, 24 x
    (QKV, attention, attn-out + residual, LayerNorm, FFN-up + GELU, FFN-down + residual, LayerNorm), pooling) of the packed
    sequences ``token_ids`` int32 [tokens] / ``cu_seqlens`` int32 [nseq + 1] -> fp32 [nseq, hidden] (not normalised).
    ``encoder_handle``: the ``rass_encoder_t*`` of a loaded encoder as an int (``HipSentenceEncoder.handle``)."""
    _req(token_ids, torch.int32, "token_ids")
    _req(cu_seqlens, torch.int32, "cu_seqlens")
    nseq = cu_seqlens.numel() - 1
    out = torch.empty((nseq, int(hidden)), dtype=torch.float32, device=token_ids.device)
    N.check("rass_encode_device",
            N.lib().rass_encode_device(ctypes.c_void_p(int(encoder_handle)), ctypes.c_void_p(token_ids.data_ptr()),
                                       ctypes.c_void_p(cu_seqlens.data_ptr()), int(nseq), int(token_ids.numel()),
                                       int(max_seqlen), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(_stream_ptr())))
    return out


# ------------------------------------------------------------------------------------------- torch.library custom ops
# BASELINE.json's north_star asks for the kernels "under PyTorch-ROCm custom ops": the stateless launchers above — search
# (scan_topk_packed, topk_merge, normalize_rows) and encoder (gemm_bf16, attention_bf16, encode) — are
# registered as torch.library ops (namespace ``rass``), so they can be called as ``torch.ops.rass.*``, traced / exported with
# their shapes known (the fake implementations below), and composed with torch code on the current stream.  They are thin:
# the arithmetic is the C ABI's (``rass_scan_topk_f32``, ``rass_topk_merge``, ``rass_normalize_rows_f32``), device tensors are
# pointer carriers.  The product's host layer (engine.py, indexer.py) keeps calling the C ABI directly.
def _register_torch_ops() -> None:
    lib = torch.library

    @lib.custom_op("rass::scan_topk_packed", mutates_args=())
    def _scan(packed: torch.Tensor, n_rows: int, queries: torch.Tensor, k: int, row_tag: Optional[torch.Tensor] = None,
              q_filter: Optional[torch.Tensor] = None, id_base: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        return scan_topk_packed(packed, n_rows, queries, k, row_tag=row_tag, q_filter=q_filter, id_base=id_base)

    @_scan.register_fake
    def _(packed, n_rows, queries, k, row_tag=None, q_filter=None, id_base=0):
        nq = queries.shape[0]
        return (queries.new_empty((nq, k), dtype=torch.float32), queries.new_empty((nq, k), dtype=torch.int64))

    @lib.custom_op("rass::topk_merge", mutates_args=())
    def _merge(list_scores: torch.Tensor, list_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return topk_merge(list_scores, list_ids)

    @_merge.register_fake
    def _(list_scores, list_ids):
        _n, nq, k = list_scores.shape
        return (list_scores.new_empty((nq, k)), list_ids.new_empty((nq, k)))

    @lib.custom_op("rass::normalize_rows", mutates_args=())
    def _norm(x: torch.Tensor, out_stride: int = 0) -> torch.Tensor:
        return normalize_rows(x, out_stride or None)

    @_norm.register_fake
    def _(x, out_stride=0):
        return x.new_empty((x.shape[0], out_stride or x.shape[1]))

    # the encoder's kernels (north_star: "MFMA bf16 GEMMs for attention/MLP ... under PyTorch-ROCm custom ops")
    @lib.custom_op("rass::gemm_bf16", mutates_args=())
    def _gemm(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, residual: Optional[torch.Tensor] = None,
              epilogue: int = 0) -> torch.Tensor:
        return gemm_bf16(x, w, bias, residual, epilogue).clone()        # (a custom op may not return a view of its scratch)

    @_gemm.register_fake
    def _(x, w, bias, residual=None, epilogue=0):
        return x.new_empty((x.shape[0], w.shape[0]))

    @lib.custom_op("rass::attention_bf16", mutates_args=())
    def _attn(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int, heads: int) -> torch.Tensor:
        return attention_bf16(qkv, cu_seqlens, max_seqlen, heads)

    @_attn.register_fake
    def _(qkv, cu_seqlens, max_seqlen, heads):
        return qkv.new_empty((qkv.shape[0], qkv.shape[1] // 3))

    @lib.custom_op("rass::attention_out_bf16", mutates_args=())
    def _attn_out(qkv: torch.Tensor, cu_seqlens: torch.Tensor, heads: int, w: torch.Tensor, bias: torch.Tensor,
                  residual: torch.Tensor) -> torch.Tensor:
        return attention_out_bf16(qkv, cu_seqlens, heads, w, bias, residual)

    @_attn_out.register_fake
    def _(qkv, cu_seqlens, heads, w, bias, residual):
        return qkv.new_empty((qkv.shape[0], w.shape[0]))

    @lib.custom_op("rass::encode", mutates_args=())
    def _encode(encoder_handle: int, token_ids: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int, hidden: int
                ) -> torch.Tensor:
        return encode(encoder_handle, token_ids, cu_seqlens, max_seqlen, hidden)

    @_encode.register_fake
    def _(encoder_handle, token_ids, cu_seqlens, max_seqlen, hidden):
        return token_ids.new_empty((cu_seqlens.shape[0] - 1, hidden), dtype=torch.float32)


try:
    _register_torch_ops()
except Exception as _e:  # pragma: no cover - an older torch without torch.library.custom_op: the plain functions remain
    import warnings
    warnings.warn(f"rassengine_amd: torch.library registration skipped ({_e})")
