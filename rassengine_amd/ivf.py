"""IVF (inverted-file) cosine index — K9 of SURVEY §8a, BASELINE cfg 5 (IVF-4096).

Division of labour:

* OFFLINE build (this file): spherical k-means on a row sample and the row -> list
  assignment.  Both are plain dense GEMMs + argmax over data already in HBM, run through
  ``torch.matmul`` (a library GEMM is the right tool for a plain GEMM; it is not on the
  query path).  With several GPUs the centroid sums are all-reduced (16.8 MB at 4096 x 1024:
  ~0.2 ms ring over xGMI), so every rank ends with identical centroids and builds lists over
  its own row shard.
* HOT PATH (HIP, behind the C ABI): ``rass_ivf_search*`` — coarse top-nprobe over the centroid
  slab with the fused flat scan, a plan kernel, and the same fused scan over the union of the
  batch's probed lists (``csrc/ivf.hip``, IVF mode of ``csrc/scan_topk.hip``).

IVF is approximate: recall@k against the flat index is measured per nprobe
(``scripts/bench_ivf.py``, ``tests/test_gpu_ivf.py``); ``nprobe = nlist`` reproduces the flat result (nprobe > 32 selects lists by a per-query score
threshold from the full centroid score matrix).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _native as N
from .engine import FlatIndex


class _DevArray:
    """A raw device pointer as a ``__cuda_array_interface__`` object (zero-copy torch view)."""

    def __init__(self, ptr: int, shape, typestr: str = "<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def _rows_chunk(index: FlatIndex, first: int, n: int, out: torch.Tensor) -> torch.Tensor:
    """Row-major fp32 rows [first, first+n) of the index's tile16 slab into ``out[:n]`` (device)."""
    N.check("rass_unpack_rows_f32",
            N.lib().rass_unpack_rows_f32(ctypes.c_void_p(index.device_rows_ptr), index.row_stride, int(first), int(n),
                                         index.dim, ctypes.c_void_p(out.data_ptr()), index.dim,
                                         ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
    return out[:n]


def _assign_chunk(x: torch.Tensor, cent_t: torch.Tensor) -> torch.Tensor:
    return torch.argmax(x @ cent_t, dim=1)


def train_centroids(index: FlatIndex, nlist: int, train_rows: int = 0, iters: int = 20, seed: int = 0,
                    group: Optional[dist.ProcessGroup] = None, chunk: int = 65536) -> torch.Tensor:
    """Spherical k-means over (a strided sample of) the index's rows; returns unit centroids
    [nlist, dim] on the GPU.  Deterministic for a given seed and shard layout."""
    dev = torch.device("cuda", index.engine.device)
    n = index.rows
    m = n if train_rows <= 0 else min(n, int(train_rows))
    if m < nlist:
        raise ValueError(f"need at least nlist={nlist} training rows, have {m}")
    index.engine.synchronize()
    step = max(1, n // m)
    buf = torch.empty((chunk, index.dim), dtype=torch.float32, device=dev)
    # strided sample, materialised once (m x dim fp32)
    sample = torch.empty((m, index.dim), dtype=torch.float32, device=dev)
    got = 0
    first = 0
    while got < m and first < n:
        cnt = min(chunk, n - first)
        rows = _rows_chunk(index, first, cnt, buf)[::step]
        take = min(rows.shape[0], m - got)
        sample[got:got + take] = rows[:take]
        got += take
        first += cnt
    sample = sample[:got]
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    world = dist.get_world_size(group) if (dist.is_initialized()) else 1
    perm = torch.randperm(got, generator=g)[:nlist].to(dev)
    cent = sample[perm].clone()
    if world > 1:  # every rank starts from rank 0's seeds
        dist.broadcast(cent, src=0, group=group)
    for _ in range(iters):
        sums = torch.zeros((nlist, index.dim), dtype=torch.float32, device=dev)
        counts = torch.zeros((nlist,), dtype=torch.float32, device=dev)
        cent_t = cent.t().contiguous()
        for a in range(0, got, chunk):
            x = sample[a:a + chunk]
            lab = _assign_chunk(x, cent_t)
            sums.index_add_(0, lab, x)
            counts.index_add_(0, lab, torch.ones_like(lab, dtype=torch.float32))
        if world > 1:
            dist.all_reduce(sums, group=group)
            dist.all_reduce(counts, group=group)
        empty = counts == 0
        new = sums / (sums.norm(dim=1, keepdim=True) + 1e-9)
        if bool(empty.any()):  # re-seed empty lists from (deterministic) sample rows
            idx = torch.nonzero(empty).flatten()
            new[idx] = sample[(idx * 7919 + 13) % got]
            if world > 1:
                dist.broadcast(new, src=0, group=group)
        cent = new
    return cent


def assign_rows(index: FlatIndex, centroids: torch.Tensor, chunk: int = 65536) -> np.ndarray:
    """List id (argmax cosine) of every row of the index; int32 host array."""
    dev = centroids.device
    n = index.rows
    index.engine.synchronize()
    buf = torch.empty((chunk, index.dim), dtype=torch.float32, device=dev)
    cent_t = centroids.t().contiguous()
    out = torch.empty((n,), dtype=torch.int32, device=dev)
    for a in range(0, n, chunk):
        cnt = min(chunk, n - a)
        out[a:a + cnt] = _assign_chunk(_rows_chunk(index, a, cnt, buf), cent_t).to(torch.int32)
    return out.cpu().numpy()


class IvfIndex:
    """IVF view of a flat index shard (its own permuted copy of the rows in HBM)."""

    def __init__(self, handle: ctypes.c_void_p, engine, dim: int):
        self._h = handle
        self.engine = engine
        self.dim = dim
        self._L = N.lib()

    @classmethod
    def build(cls, index: FlatIndex, nlist: int = 4096, train_rows: int = 0, iters: int = 20, seed: int = 0,
              group: Optional[dist.ProcessGroup] = None, centroids: Optional[torch.Tensor] = None) -> "IvfIndex":
        if centroids is None:
            centroids = train_centroids(index, nlist, train_rows, iters, seed, group)
        assign = assign_rows(index, centroids)
        c_host = np.ascontiguousarray(centroids.cpu().numpy(), dtype=np.float32)
        h = ctypes.c_void_p()
        N.check("rass_ivf_build", N.lib().rass_ivf_build(index._h, c_host.ctypes.data_as(ctypes.c_void_p), int(nlist),
                                                        assign.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)))
        ivf = cls(h, index.engine, index.dim)
        ivf.assign = assign                                   # list id of every source row (host int32)
        ivf.list_sizes = np.bincount(assign, minlength=nlist)
        return ivf

    @property
    def rows(self) -> int:
        return int(self._L.rass_ivf_rows(self._h))

    @property
    def nlist(self) -> int:
        return int(self._L.rass_ivf_nlist(self._h))

    def close(self) -> None:
        if self._h:
            self._L.rass_ivf_destroy(self._h)
            self._h = None

    def search(self, queries: np.ndarray, k: int, nprobe: int, q_filter: Optional[np.ndarray] = None
               ) -> Tuple[np.ndarray, np.ndarray, int]:
        """(scores f32 [nq,k], ids i64 [nq,k], rows touched by the fine scans)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] queries, got {q.shape}")
        f = None if q_filter is None else np.ascontiguousarray(q_filter, dtype=np.int32)
        out_s = np.empty((q.shape[0], k), dtype=np.float32)
        out_i = np.empty((q.shape[0], k), dtype=np.int64)
        scanned = ctypes.c_int64(0)
        N.check("rass_ivf_search",
                self._L.rass_ivf_search(self._h, q.ctypes.data_as(ctypes.c_void_p), q.shape[0], int(k), int(nprobe),
                                        None if f is None else f.ctypes.data_as(ctypes.c_void_p),
                                        out_s.ctypes.data_as(ctypes.c_void_p), out_i.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.byref(scanned)))
        return out_s, out_i, int(scanned.value)

    def search_device(self, d_queries_ptr: int, nq: int, k: int, nprobe: int, d_out_scores_ptr: int,
                      d_out_ids_ptr: int, d_q_filter_ptr: int = 0) -> None:
        N.check("rass_ivf_search_device",
                self._L.rass_ivf_search_device(self._h, ctypes.c_void_p(d_queries_ptr), int(nq), int(k), int(nprobe),
                                               ctypes.c_void_p(d_q_filter_ptr or 0), ctypes.c_void_p(d_out_scores_ptr),
                                               ctypes.c_void_p(d_out_ids_ptr)))


class IvfShard:
    """``LocalShard`` for ``dist.ShardedSearch``: every rank probes its own lists (shared
    centroids), the per-shard top-k meet in the same RCCL all-gather + merge as the flat path.
    Local source row ids are offset by ``id_base`` after the search."""

    def __init__(self, ivf: IvfIndex, id_base: int, nprobe: int):
        self.ivf = ivf
        self.id_base = int(id_base)
        self.nprobe = int(nprobe)
        self.device = torch.device("cuda", ivf.engine.device)
        ivf.engine.set_stream(int(torch.cuda.current_stream(self.device).cuda_stream))

    def search_local(self, queries: torch.Tensor, k: int):
        nq = queries.shape[0]
        out_s = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        self.ivf.search_device(queries.data_ptr(), nq, k, self.nprobe, out_s.data_ptr(), out_i.data_ptr())
        if self.id_base:
            out_i = torch.where(out_i >= 0, out_i + self.id_base, out_i)
        return out_s, out_i

    def merge(self, list_scores: torch.Tensor, list_ids: torch.Tensor):
        from . import ops
        return ops.topk_merge(list_scores, list_ids)
