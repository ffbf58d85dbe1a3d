"""Thin object layer over the C ABI: one ``Engine`` per process per GPU, named ``FlatIndex``
objects that live in HBM.  Host arrays are numpy; device-resident entry points take torch
tensors only as (pointer, shape) carriers — torch is plumbing for memory and streams here.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Dict, Optional, Tuple

import numpy as np

from . import _native as N


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Engine:
    """Owns one GPU's corpus slabs for the life of the process (SURVEY §8b ownership)."""

    _lock = threading.Lock()
    _singletons: Dict[Tuple[int, int], "Engine"] = {}

    def __init__(self, device: int = 0, dim: int = 1024):
        self._L = N.lib()
        h = ctypes.c_void_p()
        N.check("rass_engine_create", self._L.rass_engine_create(device, dim, ctypes.byref(h)))
        self._h = h
        self.device = device
        self.dim = dim
        self._indices: Dict[str, "FlatIndex"] = {}

    @classmethod
    def get(cls, device: int = 0, dim: int = 1024) -> "Engine":
        """Process-global engine for (device, dim): OpenSearchIndexer is built per request
        (reference app/main.py:2802), so lookups must be O(1)."""
        with cls._lock:
            eng = cls._singletons.get((device, dim))
            if eng is None:
                eng = cls._singletons[(device, dim)] = Engine(device, dim)
            return eng

    def close(self) -> None:
        if self._h:
            with Engine._lock:
                for key, eng in list(Engine._singletons.items()):
                    if eng is self:
                        del Engine._singletons[key]
            self._L.rass_engine_destroy(self._h)
            self._h = None
            self._indices.clear()

    def __del__(self):  # pragma: no cover - interpreter shutdown ordering
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int) -> None:
        """Run engine work on a caller-owned hipStream_t (0 = HIP's null stream, which is
        what ``torch.cuda.current_stream().cuda_stream`` returns for torch's default stream)."""
        N.check("rass_engine_set_stream", self._L.rass_engine_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    def reset_stream(self) -> None:
        N.check("rass_engine_reset_stream", self._L.rass_engine_reset_stream(self._h))

    @property
    def stream(self) -> int:
        """The hipStream_t (as int) engine work is enqueued on."""
        return int(self._L.rass_engine_get_stream(self._h) or 0)

    def synchronize(self) -> None:
        N.check("rass_engine_synchronize", self._L.rass_engine_synchronize(self._h))

    def kernel_timing_begin(self, max_launches: int) -> None:
        """Bracket every scan-kernel launch with a hipEvent pair on the engine stream."""
        N.check("rass_engine_kernel_timing_begin", self._L.rass_engine_kernel_timing_begin(self._h, int(max_launches)))

    def kernel_timing_end(self) -> Tuple[float, int]:
        """(summed scan-kernel milliseconds, launches) since ``kernel_timing_begin``."""
        ms = ctypes.c_double(0.0)
        n = ctypes.c_int(0)
        N.check("rass_engine_kernel_timing_end",
                self._L.rass_engine_kernel_timing_end(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    def open_index(self, name: str, capacity_rows: int = 0, dtype: str = "f32") -> "FlatIndex":
        """Look up or create the named cosine index (ensure_index_exists, app/main.py:350).  ``dtype``:
        "f32" (the parity path) or "bf16" (a bf16-only corpus: half the HBM, scores within ~1e-3)."""
        idx = self._indices.get(name)
        if idx is not None:
            return idx
        code = {"f32": N.RASS_F32, "bf16": N.RASS_BF16}[dtype]
        h = ctypes.c_void_p()
        N.check("rass_index_open",
                self._L.rass_index_open(self._h, name.encode(), code, int(capacity_rows), ctypes.byref(h)))
        idx = FlatIndex(self, name, h)
        self._indices[name] = idx
        return idx

    def search_multi(self, indices, queries: np.ndarray, k: int, q_filter: Optional[np.ndarray] = None,
                     q_filter_mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """CROSS-INDEX batch (``rass_index_search_multi``): query i is answered over ``indices[i]``; queries of
        different per-user indices share scan launches.  Returns (scores f32 [nq,k], ids i64 [nq,k]), ids being
        rows of each query's own index."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim or q.shape[0] != len(indices):
            raise ValueError(f"expected one [{self.dim}] query per index, got {q.shape} for {len(indices)} indices")
        nq = q.shape[0]
        f = None if q_filter is None else np.ascontiguousarray(q_filter, dtype=np.int32)
        m = None if q_filter_mask is None else np.ascontiguousarray(q_filter_mask, dtype=np.int32)
        if (f is not None and f.shape != (nq,)) or (m is not None and (f is None or m.shape != (nq,))):
            raise ValueError("q_filter / q_filter_mask must be one int32 per query (mask needs filter)")
        handles = (ctypes.c_void_p * nq)(*[ix._h for ix in indices])
        out_s = np.empty((nq, int(k)), dtype=np.float32)
        out_i = np.empty((nq, int(k)), dtype=np.int64)
        N.check("rass_index_search_multi",
                self._L.rass_index_search_multi(handles, _np_ptr(q), nq, int(k), _np_ptr(f), _np_ptr(m), _np_ptr(out_s),
                                                _np_ptr(out_i)))
        return out_s, out_i

    def drop_index(self, name: str) -> None:
        N.check("rass_index_drop", self._L.rass_index_drop(self._h, name.encode()))
        self._indices.pop(name, None)

    def load_index(self, name: str, path: str) -> "FlatIndex":
        h = ctypes.c_void_p()
        N.check("rass_index_load", self._L.rass_index_load(self._h, name.encode(), path.encode(), ctypes.byref(h)))
        idx = FlatIndex(self, name, h)
        self._indices[name] = idx
        return idx


class FlatIndex:
    """Row-major fp32 cosine index resident in HBM; row ids are insertion ordinals."""

    def __init__(self, engine: Engine, name: str, handle: ctypes.c_void_p):
        self.engine = engine
        self.name = name
        self._h = handle
        self._L = engine._L

    # ---- bookkeeping
    @property
    def count(self) -> int:
        """Live rows (OpenSearchIndexer.has_any_data's count, app/main.py:1475)."""
        return int(self._L.rass_index_count(self._h))

    @property
    def rows(self) -> int:
        return int(self._L.rass_index_rows(self._h))

    @property
    def dim(self) -> int:
        return int(self._L.rass_index_dim(self._h))

    @property
    def row_stride(self) -> int:
        return int(self._L.rass_index_row_stride(self._h))

    MULTI_MAX_TILES = 65536      # rass_index_search_multi: 32-row tiles per cross-index batch (kMultiMaxItems)

    @property
    def dtype(self) -> str:
        return "bf16" if int(self._L.rass_index_dtype(self._h)) == 1 else "f32"

    @property
    def has_global_ids(self) -> bool:
        return int(self._L.rass_index_has_global_ids(self._h)) != 0

    @property
    def multi_tiles(self) -> int:
        """Tiles this index would take of a cross-index batch's budget; 0 = it cannot join one (bf16 corpus,
        caller-assigned ids, or too large on its own): search it through its own batcher."""
        if self.dtype != "f32" or self.has_global_ids or self.row_stride > 1024:
            return 0
        tiles = max(1, (self.rows + 31) // 32)        # an empty index may join too (it answers with padding)
        return tiles if tiles <= self.MULTI_MAX_TILES // 2 else 0

    @property
    def device_rows_ptr(self) -> int:
        """Device pointer of the tile16 slab (invalidated by growth)."""
        return int(self._L.rass_index_device_rows(self._h) or 0)

    @property
    def device_tags_ptr(self) -> int:
        return int(self._L.rass_index_device_tags(self._h) or 0)

    # ---- write path
    def add(self, vecs: np.ndarray, tags: Optional[np.ndarray] = None, normalize: bool = True,
            first_global_id: int = -1) -> int:
        """Append rows (host fp32 [n, dim]); returns the ORDINAL of the first appended row.
        ``first_global_id`` >= 0: searches report ``first_global_id + i`` for row i of this batch instead
        of its ordinal (a shard of a multi-GPU index, ``rass_index_add_ex``)."""
        v = np.ascontiguousarray(vecs, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}] vectors, got {v.shape}")
        t = None
        if tags is not None:
            t = np.ascontiguousarray(tags, dtype=np.int32)
            if t.shape != (v.shape[0],):
                raise ValueError("tags must be one int32 per row")
        first = ctypes.c_int64(-1)
        N.check("rass_index_add_ex", self._L.rass_index_add_ex(self._h, _np_ptr(v), _np_ptr(t), v.shape[0],
                                                              1 if normalize else 0, int(first_global_id), 0,
                                                              ctypes.byref(first)))
        return int(first.value)

    def add_device(self, d_vecs_ptr: int, n: int, d_tags_ptr: int = 0, normalize: bool = True) -> int:
        first = ctypes.c_int64(-1)
        N.check("rass_index_add_device",
                self._L.rass_index_add_device(self._h, ctypes.c_void_p(d_vecs_ptr), ctypes.c_void_p(d_tags_ptr or 0),
                                              int(n), 1 if normalize else 0, ctypes.byref(first)))
        return int(first.value)

    def fill_synthetic(self, n: int, seed: int, row_id_base: int = 0) -> None:
        N.check("rass_index_fill_synthetic",
                self._L.rass_index_fill_synthetic(self._h, int(n), ctypes.c_uint64(seed), int(row_id_base)))

    def delete(self, row: int) -> None:
        N.check("rass_index_delete", self._L.rass_index_delete(self._h, int(row)))

    def get_row(self, row: int) -> np.ndarray:
        out = np.empty(self.dim, dtype=np.float32)
        N.check("rass_index_get_row", self._L.rass_index_get_row(self._h, int(row), _np_ptr(out)))
        return out

    def get_rows(self, first_row: int, n: int) -> np.ndarray:
        """Stored (normalised) rows [first_row, first_row+n) as fp32 [n, dim]."""
        out = np.empty((int(n), self.dim), dtype=np.float32)
        N.check("rass_index_get_rows", self._L.rass_index_get_rows(self._h, int(first_row), int(n), _np_ptr(out)))
        return out

    PREFILTER_MODES = {False: 0, None: 0, 0: 0, "off": 0, True: 1, 1: 1, "bf16": 1, 2: 2, "int8": 2}

    def set_prefilter(self, enable=True) -> None:
        """Candidate scan over a reduced copy of the slab + exact fp32 re-rank; off by default.
        ``enable``: False / "off", True / "bf16" (half the bytes per pass) or "int8" (a quarter)."""
        try:
            mode = self.PREFILTER_MODES[enable]
        except (KeyError, TypeError):
            raise ValueError(f"prefilter mode must be one of off / bf16 / int8, not {enable!r}") from None
        N.check("rass_index_set_prefilter", self._L.rass_index_set_prefilter(self._h, mode))

    @property
    def prefilter(self) -> bool:
        return bool(self._L.rass_index_get_prefilter(self._h))

    @property
    def prefilter_mode(self) -> str:
        return ("off", "bf16", "int8")[int(self._L.rass_index_get_prefilter(self._h))]

    def candidates_device(self, d_queries, q_filter=None):
        """The active prefilter mode's candidate lists BEFORE the exact re-rank, for <= 32 queries on the device (a torch
        CUDA fp32 tensor [nq, dim]): (scores f32 [nq, 32], LOCAL rows i64 [nq, 32]) as torch tensors.  A parity hook: the
        int8 path is integer work and is compared with the oracle bit for bit (tests/test_gpu_prefilter_int8.py)."""
        import torch
        nq = int(d_queries.shape[0])
        assert d_queries.is_cuda and d_queries.dtype == torch.float32 and d_queries.is_contiguous() and d_queries.shape[1] == self.dim
        s = torch.empty((nq, 32), dtype=torch.float32, device=d_queries.device)
        r = torch.empty((nq, 32), dtype=torch.int64, device=d_queries.device)
        f = None
        if q_filter is not None:
            f = torch.as_tensor(np.ascontiguousarray(q_filter, dtype=np.int32)).to(d_queries.device)
        torch.cuda.current_stream().synchronize()     # the engine works on its own stream
        N.check("rass_index_candidates_device",
                self._L.rass_index_candidates_device(self._h, ctypes.c_void_p(d_queries.data_ptr()), nq,
                                                     ctypes.c_void_p(f.data_ptr()) if f is not None else None,
                                                     ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(r.data_ptr())))
        self.engine.synchronize()
        return s, r

    def save(self, path: str) -> None:
        N.check("rass_index_save", self._L.rass_index_save(self._h, path.encode()))

    # ---- read path
    def search(self, queries: np.ndarray, k: int, q_filter: Optional[np.ndarray] = None,
               q_filter_mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """Exact cosine top-k.  Returns (scores f32 [nq,k], ids i64 [nq,k]); raw cosine, best
        first, ties by id ascending, (-inf, -1) padding.  ``q_filter`` restricts query q to rows whose
        tag equals it (-1 = no filter); with ``q_filter_mask`` to rows with ``(tag & mask) == filter``.
        Any k >= 1: k > 32 is served exactly in passes of 32 (``rass_index_search_ex``).  Thread-safe."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] queries, got {q.shape}")
        f = m = None
        if q_filter is not None:
            f = np.ascontiguousarray(q_filter, dtype=np.int32)
            if f.shape != (q.shape[0],):
                raise ValueError("q_filter must be one int32 per query")
        if q_filter_mask is not None:
            if f is None:
                raise ValueError("q_filter_mask needs q_filter")
            m = np.ascontiguousarray(q_filter_mask, dtype=np.int32)
            if m.shape != (q.shape[0],):
                raise ValueError("q_filter_mask must be one int32 per query")
        k = int(k)
        out_s = np.empty((q.shape[0], k), dtype=np.float32)
        out_i = np.empty((q.shape[0], k), dtype=np.int64)
        N.check("rass_index_search_ex", self._L.rass_index_search_ex(self._h, _np_ptr(q), q.shape[0], k, _np_ptr(f),
                                                                    _np_ptr(m), _np_ptr(out_s), _np_ptr(out_i)))
        return out_s, out_i

    def search_device(self, d_queries_ptr: int, nq: int, k: int, d_out_scores_ptr: int, d_out_ids_ptr: int,
                      id_base: int = 0, d_q_filter_ptr: int = 0, d_q_filter_mask_ptr: int = 0) -> None:
        """Async, device-resident variant (multi-GPU path, benchmark); nq <= 32."""
        N.check("rass_index_search_device_ex",
                self._L.rass_index_search_device_ex(self._h, ctypes.c_void_p(d_queries_ptr), int(nq), int(k),
                                                    ctypes.c_void_p(d_q_filter_ptr or 0),
                                                    ctypes.c_void_p(d_q_filter_mask_ptr or 0), int(id_base),
                                                    ctypes.c_void_p(d_out_scores_ptr), ctypes.c_void_p(d_out_ids_ptr)))

    def search_device_after(self, d_queries_ptr: int, nq: int, k: int, d_after_score_ptr: int, d_after_row_ptr: int,
                            d_out_scores_ptr: int, d_out_ids_ptr: int, d_q_filter_ptr: int = 0,
                            d_q_filter_mask_ptr: int = 0) -> None:
        """One continuation pass (``rass_index_search_device_after``): only rows strictly after (after_score[q],
        after_row[q]) in (score desc, row asc) rank for query q.  Async, device-resident, nq <= 32, k <= 32."""
        N.check("rass_index_search_device_after",
                self._L.rass_index_search_device_after(self._h, ctypes.c_void_p(d_queries_ptr), int(nq), int(k),
                                                       ctypes.c_void_p(d_q_filter_ptr or 0),
                                                       ctypes.c_void_p(d_q_filter_mask_ptr or 0),
                                                       ctypes.c_void_p(d_after_score_ptr), ctypes.c_void_p(d_after_row_ptr),
                                                       ctypes.c_void_p(d_out_scores_ptr), ctypes.c_void_p(d_out_ids_ptr)))

    def search_device_batch(self, d_queries_ptr: int, nq: int, k: int, d_out_scores_ptr: int, d_out_ids_ptr: int,
                            id_base: int = 0, d_q_filter_ptr: int = 0, out_scores_group_stride: int = 0,
                            out_ids_group_stride: int = 0) -> None:
        """Async, device-resident, MANY launch groups per call (nq <= 4096, k <= 32): bit-identical to
        ``search_device`` on consecutive groups of 32 queries, with one normalise and one merge launch for the whole
        batch.  Group g's results go to out + g * group_stride (elements; 0 = contiguous [nq][k])."""
        N.check("rass_index_search_device_batch",
                self._L.rass_index_search_device_batch(self._h, ctypes.c_void_p(d_queries_ptr), int(nq), int(k),
                                                       ctypes.c_void_p(d_q_filter_ptr or 0), int(id_base),
                                                       ctypes.c_void_p(d_out_scores_ptr), ctypes.c_void_p(d_out_ids_ptr),
                                                       int(out_scores_group_stride), int(out_ids_group_stride)))


class HipTimer:
    """hipEvent pair on an explicit stream (rass_timer_*)."""

    def __init__(self):
        self._L = N.lib()
        h = ctypes.c_void_p()
        N.check("rass_timer_create", self._L.rass_timer_create(ctypes.byref(h)))
        self._h = h

    def start(self, stream_ptr: int = 0) -> None:
        N.check("rass_timer_start", self._L.rass_timer_start(self._h, ctypes.c_void_p(stream_ptr or 0)))

    def stop(self, stream_ptr: int = 0) -> None:
        N.check("rass_timer_stop", self._L.rass_timer_stop(self._h, ctypes.c_void_p(stream_ptr or 0)))

    def elapsed_ms(self) -> float:
        ms = ctypes.c_float(0)
        N.check("rass_timer_elapsed_ms", self._L.rass_timer_elapsed_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def __del__(self):  # pragma: no cover
        try:
            if self._h:
                self._L.rass_timer_destroy(self._h)
                self._h = None
        except Exception:
            pass


def scan_kernel_name(dim: int, nq: int) -> str:
    return N.lib().rass_scan_kernel_name(int(dim), int(nq)).decode()
