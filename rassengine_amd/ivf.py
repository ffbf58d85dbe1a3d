"""IVF (inverted-file) cosine index — K9 of SURVEY §8a, BASELINE cfg 5 (IVF-4096).

Division of labour:

* BUILD (offline): spherical k-means on a strided sample of the rows and the row -> list assignment.  The two
  O(rows) steps are HIP kernels behind the C ABI, working on the rows where they already are (the index's tile16
  slab in HBM): ``rass_kmeans_assign`` (exact fp32 MFMA GEMM rows x centroids^T with a row-argmax epilogue,
  ``csrc/kmeans.hip``) and ``rass_kmeans_accumulate`` (centroid sums / counts, fp32 atomics).  What stays here is
  O(nlist x dim): the all-reduce of sums and counts across ranks (16.8 MB at 4096 x 1024: ~0.2 ms ring over
  xGMI, so every rank ends with identical centroids and builds lists over its own row shard), the normalisation
  of the new centroids (``rass_normalize_rows_f32``) and the re-seeding of empty lists.  No library GEMM.
* HOT PATH (HIP, behind the C ABI): ``rass_ivf_search*`` — coarse top-nprobe over the centroid slab with the
  fused flat scan, a plan kernel, and the same fused scan over the union of the batch's probed lists
  (``csrc/ivf.hip``, IVF mode of ``csrc/scan_topk.hip``).

IVF is approximate BY CONSTRUCTION, with exact semantics: a query's result is the brute-force top-k restricted to
the rows of its nprobe best-scoring lists (``tests/test_gpu_cfg5.py`` pins that against the oracle); recall@k
against the flat index is therefore a property of the data and of nprobe, measured per configuration
(``scripts/bench_ivf.py``), and ``nprobe = nlist`` reproduces the flat result bit for bit.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _native as N
from . import ops
from .engine import FlatIndex

BLOCK_ROWS = 32  # the k-means kernels work on 32-row blocks of the slab


class _engine_on_torch_stream:
    """Run the engine's kernels on torch's current stream for the duration (they are ordered with the torch
    ops around them without host synchronisation), then put its stream back."""

    def __init__(self, index: FlatIndex):
        self.engine = index.engine
        self.prev = None

    def __enter__(self):
        self.prev = self.engine.stream
        dev = torch.device("cuda", self.engine.device)
        self.engine.set_stream(int(torch.cuda.current_stream(dev).cuda_stream))
        return self

    def __exit__(self, *exc):
        torch.cuda.current_stream(torch.device("cuda", self.engine.device)).synchronize()
        self.engine.set_stream(self.prev)
        return False


def _pack_centroids(cent: torch.Tensor, row_stride: int) -> torch.Tensor:
    """Row-major [nlist, dim] -> normalised tile16 slab (what the assign kernel and the probe stream)."""
    return ops.pack_rows(cent.contiguous(), normalize=True, row_stride=row_stride)


def kmeans_assign(index: FlatIndex, centroids: torch.Tensor, first_block: int = 0, block_step: int = 1,
                  n_blocks: Optional[int] = None, with_best: bool = False):
    """List id (arg max cosine, exact fp32) of the rows of the processed 32-row blocks; device int32
    [n_blocks * 32] (entries of rows past ``index.rows`` are meaningless).  Async on torch's current stream
    when called inside ``_engine_on_torch_stream``."""
    dev = centroids.device
    if n_blocks is None:
        n_blocks = max(0, -(-(index.rows - first_block * BLOCK_ROWS) // (BLOCK_ROWS * block_step)))
    slab = _pack_centroids(centroids, index.row_stride)
    assign = torch.empty((n_blocks * BLOCK_ROWS,), dtype=torch.int32, device=dev)
    best = torch.empty((n_blocks * BLOCK_ROWS,), dtype=torch.float32, device=dev) if with_best else None
    N.check("rass_kmeans_assign",
            N.lib().rass_kmeans_assign(index._h, int(first_block), int(block_step), int(n_blocks),
                                       ctypes.c_void_p(slab.data_ptr()), int(centroids.shape[0]),
                                       ctypes.c_void_p(assign.data_ptr()),
                                       ctypes.c_void_p(best.data_ptr()) if best is not None else None))
    return (assign, best, slab) if with_best else (assign, slab)


def kmeans_accumulate(index: FlatIndex, assign: torch.Tensor, nlist: int, first_block: int, block_step: int,
                      n_blocks: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(sums [nlist, dim] fp32, counts [nlist] fp32) of the processed rows by list."""
    dev = assign.device
    sums = torch.zeros((nlist, index.dim), dtype=torch.float32, device=dev)
    counts = torch.zeros((nlist,), dtype=torch.float32, device=dev)
    N.check("rass_kmeans_accumulate",
            N.lib().rass_kmeans_accumulate(index._h, int(first_block), int(block_step), int(n_blocks),
                                           ctypes.c_void_p(assign.data_ptr()), ctypes.c_void_p(sums.data_ptr()),
                                           ctypes.c_void_p(counts.data_ptr()), int(nlist)))
    return sums, counts


def _sample_row(j, step: int):
    """Global row of entry j of the strided sample (every ``step``-th 32-row block)."""
    return (j // BLOCK_ROWS) * step * BLOCK_ROWS + (j % BLOCK_ROWS)


def _gather_rows(index: FlatIndex, rows, dev) -> torch.Tensor:
    """Row-major fp32 copies of the given (scattered) rows of the index's slab: one ``rass_gather_rows_f32`` launch."""
    ids = torch.as_tensor(list(rows), dtype=torch.int64).to(dev)
    out = torch.zeros((ids.numel(), index.dim), dtype=torch.float32, device=dev)
    N.check("rass_gather_rows_f32",
            N.lib().rass_gather_rows_f32(ctypes.c_void_p(index.device_rows_ptr), index.row_stride, int(index.rows),
                                         ctypes.c_void_p(ids.data_ptr()), int(ids.numel()), index.dim,
                                         ctypes.c_void_p(out.data_ptr()), index.dim,
                                         ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
    return out


def _repair(index: FlatIndex, cent: torch.Tensor, counts: torch.Tensor, best: torch.Tensor, step: int, n_blocks: int,
            g: torch.Generator, dev, dup_cos: float = 0.8) -> Tuple[torch.Tensor, int]:
    """One repair of a set of list means between two Lloyd iterations: lists whose means nearly coincide (cosine above
    ``dup_cos``: two seeds fell into ONE cluster and split it) are merged — the smaller list of each such pair gives up its
    centroid — and the freed centroids are re-seeded at sample rows that no mean is close to, drawn with probability
    proportional to (1 - best cosine)^2.  Means, not rows, are compared: at sigma = 2 two rows of one cluster have cosine
    0.2 but the means of two lists that split a cluster 0.97, and a row's cosine to its cluster's mean is 0.45 against
    ~0.1 for a row whose cluster has no list.  The nearest-other-mean search is the engine's own flat scan over a scratch
    index holding the nlist means (k = 2).  Returns (centroids, number re-seeded)."""
    n, nlist = index.rows, cent.shape[0]
    eng = index.engine
    name = f"__kmeans_repair_{id(cent)}"
    scratch = eng.open_index(name, capacity_rows=nlist)
    try:
        scratch.add_device(cent.contiguous().data_ptr(), nlist, normalize=True)
        out_s = torch.empty((nlist, 2), dtype=torch.float32, device=dev)
        out_i = torch.empty((nlist, 2), dtype=torch.int64, device=dev)
        for a0 in range(0, nlist, 32):
            b0 = min(32, nlist - a0)
            scratch.search_device(cent[a0:a0 + b0].contiguous().data_ptr(), b0, 2, out_s[a0:a0 + b0].data_ptr(),
                                  out_i[a0:a0 + b0].data_ptr())
        torch.cuda.current_stream(dev).synchronize()
    finally:
        eng.drop_index(name)
    me = torch.arange(nlist, device=dev)
    other_is_second = out_i[:, 0] == me
    nn_i = torch.where(other_is_second, out_i[:, 1], out_i[:, 0])
    nn_s = torch.where(other_is_second, out_s[:, 1], out_s[:, 0])
    cnt = counts.to(dev)
    # list i yields to its near-duplicate j when it is the smaller of the two (ties: the higher index yields)
    yields = (nn_s > dup_cos) & (nn_i >= 0) & ((cnt < cnt[nn_i.clamp(min=0)]) | ((cnt == cnt[nn_i.clamp(min=0)]) & (me > nn_i)))
    freed = torch.nonzero(yields).flatten()
    if freed.numel() == 0:
        return cent, 0
    m_sample = min(n_blocks * BLOCK_ROWS, n, 1 << 24)
    rows_global = _sample_row(torch.arange(m_sample, device=dev), step)
    w = torch.clamp(1.0 - best[:m_sample].float(), min=0.0) ** 2
    w = torch.where(rows_global < n, w, torch.zeros_like(w))
    idx = torch.multinomial(w.cpu().double(), int(freed.numel()), replacement=False, generator=g)
    cent = cent.clone()
    cent[freed] = _gather_rows(index, torch.clamp(_sample_row(idx, step), max=n - 1).tolist(), dev)
    return cent, int(freed.numel())


def _lloyd(index: FlatIndex, cent: torch.Tensor, iters: int, step: int, n_blocks: int, g: torch.Generator, repair: bool,
           group: Optional[dist.ProcessGroup]) -> Tuple[torch.Tensor, torch.Tensor]:
    """``iters`` spherical Lloyd iterations over the strided sample, starting from ``cent``; returns (unit means, the sizes
    their lists had in the last assignment), identical on every rank.  Called inside ``_engine_on_torch_stream``."""
    dev = cent.device
    n, nlist = index.rows, cent.shape[0]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    one = torch.empty((1, index.dim), dtype=torch.float32, device=dev)
    counts = torch.zeros((nlist,), dtype=torch.float32, device=dev)
    for it in range(iters):
        assign, _slab = kmeans_assign(index, cent, 0, step, n_blocks)
        sums, counts = kmeans_accumulate(index, assign, nlist, 0, step, n_blocks)
        if world > 1:
            dist.all_reduce(sums, group=group)
            dist.all_reduce(counts, group=group)
        new = ops.normalize_rows(sums)                      # sums / (||sums|| + 1e-9)
        empty = counts == 0
        if bool(empty.any()):  # re-seed empty lists from (deterministic) sample rows; rank 0's win
            idx = torch.nonzero(empty).flatten()
            for li in idx.tolist():
                r = min(n - 1, ((li * 7919 + 13) % n_blocks) * step * BLOCK_ROWS + (li % BLOCK_ROWS))
                new[li] = _rows_chunk(index, int(r), 1, one)[0]
            if world > 1:
                dist.broadcast(new, src=0, group=group)
        cent = new
        if repair and it % 2 == 1 and it < (2 * iters) // 3:
            # rank 0's sample decides what is merged and where the freed centroids go; everyone gets its result
            if rank == 0:
                _a, best, _slab = kmeans_assign(index, cent, 0, step, n_blocks, with_best=True)
                cent, _n_fixed = _repair(index, cent, counts, best, step, n_blocks, g, dev)
            if world > 1:
                dist.broadcast(cent, src=0, group=group)
    return cent, counts


def _group_means(fine: torch.Tensor, counts: torch.Tensor, nlist: int, seed: int, iters: int = 15) -> torch.Tensor:
    """The second level of two-level training: spherical k-means over the K' FINE list means, each weighted by its list's
    size, into ``nlist`` groups; returns the groups' unit means (= the means of their rows, up to the fine lists' own
    spread).  Seeds by D^2 sampling over the fine means (k-means++ — affordable and effective HERE: K' points, and means are
    sharp where rows are not: two fine means of one cluster have cosine ~0.97, of two clusters ~0), so that fine lists of
    one cluster meet in one group and every group starts in a cluster of its own.  K' x nlist x dim flops per iteration:
    nothing next to one assignment of the rows."""
    K, dev = fine.shape[0], fine.device
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed) + 7919)
    w = counts.clamp(min=0).float().to(dev)
    w = torch.where(w > 0, w, torch.zeros_like(w))
    first = int(torch.multinomial(w + 1e-9, 1, generator=g))
    picks = [first]
    d = (1.0 - fine @ fine[first]).clamp(min=0.0)
    for _ in range(1, nlist):
        p = w * d * d
        if float(p.sum()) <= 0:          # fewer distinct means than groups: any unused mean will do
            p = torch.ones_like(p)
            p[torch.tensor(picks, device=dev)] = 0
        nxt = int(torch.multinomial(p, 1, generator=g))
        picks.append(nxt)
        d = torch.minimum(d, (1.0 - fine @ fine[nxt]).clamp(min=0.0))
    cent = fine[torch.tensor(picks, device=dev)].clone()
    for it in range(iters + 1):
        grp = (fine @ cent.T).argmax(dim=1)
        sums = torch.zeros_like(cent).index_add_(0, grp, fine * w[:, None])
        norm = sums.norm(dim=1)
        ok = norm > 0                     # a group that lost its members keeps its mean
        cent[ok] = sums[ok] / norm[ok][:, None]
    return cent


def train_centroids(index: FlatIndex, nlist: int, train_rows: int = 0, iters: int = 20, seed: int = 0,
                    group: Optional[dist.ProcessGroup] = None, seeding: str = "repair", fine_factor: int = 4) -> torch.Tensor:
    """Spherical k-means over a strided sample of the index's 32-row blocks (every ``step``-th block, about
    ``train_rows`` rows; 0 = all rows for one level, 64 rows per fine list for two); returns unit centroids [nlist, dim] on
    the GPU, identical on every rank.

    ``fine_factor`` > 1 (default 4) trains in TWO LEVELS: K' = fine_factor x nlist fine lists from K' random sample rows
    (plain Lloyd, ``iters`` iterations), then the fine means — each weighted by its list's size — are grouped into nlist
    groups by a k-means of their own, and the groups' means are the centroids.  Why: with more natural clusters than lists
    (SURVEY §8d's corpus has 8 192 for IVF-4096) Lloyd from nlist seeds leaves the clusters without a seed scattered over
    all lists and cannot gather them however long it runs (sigma = 2: recall@10 0.69 after 10 iterations, 0.79 after 300);
    over-clustered, nearly every cluster has a fine list or two of its own, and whole fine lists — not rows — are then
    dealt to the coarse lists (sigma = 2: 0.994 at nprobe 1, 1.0 from nprobe 4, for 2.6 s instead of 1.5;
    profiles/r03_ivf4096_2M_sigma2_two_level.json.txt).  K' is capped at 65 536 (the assign kernel's limit) and at a
    sixteenth of the sample; below 2 x nlist the training falls back to one level.

    One level (``fine_factor`` <= 1): seeds are nlist distinct sample rows; ``seeding`` "random" = nothing else (rounds
    1-2), "repair" = in the first two thirds of the iterations every other one is followed by ``_repair``: near-duplicate
    list means are merged and the freed centroids re-seeded where no mean is close (what k-means++ aims at, done on means
    instead of rows)."""
    if seeding not in ("random", "repair"):
        raise ValueError("seeding must be 'random' or 'repair'")
    dev = torch.device("cuda", index.engine.device)
    n = index.rows
    total_blocks = -(-n // BLOCK_ROWS)
    m = n if train_rows <= 0 else min(n, int(train_rows))
    if train_rows <= 0 and fine_factor and fine_factor > 1:
        # "all rows" with two levels means a strided sample of 64 rows per fine list (1 M rows for IVF-4096: what the
        # profiles were measured on): the fine assignment costs K' x dim x 2 flops per row and iteration
        m = min(n, 64 * min(int(fine_factor) * nlist, 65536))
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if m < nlist and world == 1:
        raise ValueError(f"need at least nlist={nlist} training rows, have {m}")
    step = max(1, total_blocks // max(1, -(-max(m, 1) // BLOCK_ROWS)))
    n_blocks = -(-total_blocks // step)
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sample = min(n_blocks * BLOCK_ROWS, n)
    # every rank must train the SAME number of first-level lists in the same mode (the broadcast of the seeds and the
    # all-reduce of sums / counts carry [k_first, dim] tensors): size them from the SMALLEST rank's sample, not from the
    # rank-local row count (shards hold unequal rows)
    sample_all = sample
    if world > 1:
        t = torch.tensor([sample], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        sample_all = int(t.item())
    if sample_all < nlist:
        raise ValueError(f"need at least nlist={nlist} training rows on every rank, the smallest sample has {sample_all}")
    k_fine = min(int(fine_factor) * nlist, 65536, sample_all // 16) if fine_factor and fine_factor > 1 else 0
    two_level = k_fine >= 2 * nlist
    k_first = k_fine if two_level else nlist
    repair = (not two_level) and seeding == "repair" and nlist >= 8 and sample_all >= 2 * nlist
    with _engine_on_torch_stream(index):
        # seeds: distinct sample rows (deterministic for a seed); every rank starts from rank 0's
        pick = torch.randperm(sample, generator=g)[:k_first]
        cent = _gather_rows(index, torch.clamp(_sample_row(pick, step), max=n - 1).tolist(), dev)
        if world > 1:
            dist.broadcast(cent, src=0, group=group)
        cent, counts = _lloyd(index, cent, iters, step, n_blocks, g, repair, group)
        if two_level:
            # rank 0 groups the fine means (float atomics in the weighted sums: one rank decides, everyone gets its result)
            coarse = _group_means(cent, counts, nlist, seed) if rank == 0 else torch.empty((nlist, index.dim), dtype=torch.float32, device=dev)
            if world > 1:
                dist.broadcast(coarse, src=0, group=group)
            cent = coarse
    return cent


def assign_rows(index: FlatIndex, centroids: torch.Tensor) -> np.ndarray:
    """List id (arg max cosine) of every row of the index; int32 host array."""
    with _engine_on_torch_stream(index):
        assign, _slab = kmeans_assign(index, centroids)
        out = assign[:index.rows].cpu().numpy()
    return out


def _rows_chunk(index: FlatIndex, first: int, n: int, out: torch.Tensor) -> torch.Tensor:
    """Row-major fp32 rows [first, first+n) of the index's tile16 slab into ``out[:n]`` (device)."""
    N.check("rass_unpack_rows_f32",
            N.lib().rass_unpack_rows_f32(ctypes.c_void_p(index.device_rows_ptr), index.row_stride, int(first), int(n),
                                         index.dim, ctypes.c_void_p(out.data_ptr()), index.dim,
                                         ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
    return out[:n]


SLAB_DTYPES = {"f32": 0, "bf16": 1, "int8": 2}     # rass_dtype values of rass_ivf_build_ex's slab_dtype


class IvfIndex:
    """IVF view of a flat index shard (its own permuted copy of the rows in HBM)."""

    def __init__(self, handle: ctypes.c_void_p, engine, dim: int):
        self._h = handle
        self.engine = engine
        self.dim = dim
        self._L = N.lib()

    @classmethod
    def build(cls, index: FlatIndex, nlist: int = 4096, train_rows: int = 0, iters: int = 20, seed: int = 0,
              group: Optional[dist.ProcessGroup] = None, centroids: Optional[torch.Tensor] = None,
              dtype: str = "f32", assign: Optional[np.ndarray] = None, n_rows: int = -1) -> "IvfIndex":
        """``dtype="bf16"``: the IVF keeps its list-ordered copy of the rows in bf16 (``rass_ivf_build_ex``): half the bytes
        per probed row, the scores of a flat bf16 index over the same rows.  ``dtype="int8"``: an fp32 copy PLUS its
        per-row-scaled int8 copy — the fine scan reads the int8 bytes (a quarter per probed row) for 32 candidates per query,
        which are rescored exactly from the fp32 copy: the fp32 IVF's scores, k <= 16.  The source index stays fp32."""
        if dtype not in SLAB_DTYPES:
            raise ValueError(f"dtype must be one of {sorted(SLAB_DTYPES)}, got {dtype!r}")
        if centroids is None:
            centroids = train_centroids(index, nlist, train_rows, iters, seed, group)
        if assign is None:
            assign = assign_rows(index, centroids)
        else:   # the caller's lists (e.g. two-level training: a row follows its fine list's group)
            assign = np.ascontiguousarray(assign, dtype=np.int32)
            if assign.ndim != 1 or assign.shape[0] < (index.rows if n_rows < 0 else n_rows):
                raise ValueError(f"assign must hold one list id per row ({index.rows}), got {assign.shape}")
        c_host = np.ascontiguousarray(centroids.cpu().numpy(), dtype=np.float32)
        h = ctypes.c_void_p()
        # n_rows >= 0: an IVF over the first n_rows source rows only (the rest is the flat delta of an IvfBackedIndex)
        N.check("rass_ivf_build_prefix",
                N.lib().rass_ivf_build_prefix(index._h, c_host.ctypes.data_as(ctypes.c_void_p), int(nlist),
                                              assign.ctypes.data_as(ctypes.c_void_p), SLAB_DTYPES[dtype],
                                              int(n_rows), ctypes.byref(h)))
        ivf = cls(h, index.engine, index.dim)
        ivf.assign = assign[:ivf.covered_rows]                # list id of every covered source row (host int32)
        ivf.centroids = centroids                             # [nlist, dim] (device) as trained, before the engine's normalise
        ivf.list_sizes = np.bincount(ivf.assign, minlength=nlist)
        return ivf

    def save(self, path: str) -> None:
        """The whole device state of this IVF shard (``rass_ivf_save``), written next to a temporary name and
        renamed into place."""
        import os
        tmp = path + ".tmp"
        N.check("rass_ivf_save", self._L.rass_ivf_save(self._h, tmp.encode()))
        os.replace(tmp, path)

    @classmethod
    def load(cls, engine, path: str) -> "IvfIndex":
        h = ctypes.c_void_p()
        N.check("rass_ivf_load", N.lib().rass_ivf_load(engine._h, path.encode(), ctypes.byref(h)))
        return cls(h, engine, engine.dim)

    @property
    def rows(self) -> int:
        return int(self._L.rass_ivf_rows(self._h))

    @property
    def nlist(self) -> int:
        return int(self._L.rass_ivf_nlist(self._h))

    @property
    def covered_rows(self) -> int:
        """Source rows [0, covered) are in the IVF (live or tombstoned); later rows are somebody's flat delta."""
        return int(self._L.rass_ivf_covered_rows(self._h))

    def delete(self, src_row: int) -> None:
        """Tombstone a source row inside the IVF's slab (no-op when it is not covered)."""
        N.check("rass_ivf_delete", self._L.rass_ivf_delete(self._h, int(src_row)))

    def search_device_batch(self, d_queries_ptr: int, nq: int, k: int, nprobe: int, d_out_scores_ptr: int,
                            d_out_ids_ptr: int, d_q_filter_ptr: int = 0, d_scanned_ptr: int = 0) -> None:
        """``rass_ivf_search_device_batch``: up to 1 024 queries per call, bit-identical to ``search_device`` on consecutive
        groups of 32, with one normalise / coarse / plan / merge launch for the whole batch (4 + G launches, not 5 G)."""
        N.check("rass_ivf_search_device_batch",
                self._L.rass_ivf_search_device_batch(self._h, ctypes.c_void_p(d_queries_ptr), int(nq), int(k), int(nprobe),
                                                     ctypes.c_void_p(d_q_filter_ptr or 0), ctypes.c_void_p(d_out_scores_ptr),
                                                     ctypes.c_void_p(d_out_ids_ptr), ctypes.c_void_p(d_scanned_ptr or 0)))

    def search_delta(self, flat: FlatIndex, queries: np.ndarray, k: int, nprobe: int, q_filter: Optional[np.ndarray] = None,
                     q_filter_mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray, int]:
        """``rass_ivf_search_delta``: the probe of this IVF + the exact scan of ``flat``'s rows behind ``covered_rows``,
        merged.  (scores, ids, rows touched)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] queries, got {q.shape}")
        f = None if q_filter is None else np.ascontiguousarray(q_filter, dtype=np.int32)
        m = None if q_filter_mask is None else np.ascontiguousarray(q_filter_mask, dtype=np.int32)
        if (f is not None and f.shape != (q.shape[0],)) or (m is not None and (f is None or m.shape != (q.shape[0],))):
            raise ValueError("q_filter / q_filter_mask must be one int32 per query (mask needs filter)")
        out_s = np.empty((q.shape[0], int(k)), dtype=np.float32)
        out_i = np.empty((q.shape[0], int(k)), dtype=np.int64)
        scanned = ctypes.c_int64(0)
        N.check("rass_ivf_search_delta",
                self._L.rass_ivf_search_delta(self._h, flat._h, q.ctypes.data_as(ctypes.c_void_p), q.shape[0], int(k),
                                              int(nprobe), None if f is None else f.ctypes.data_as(ctypes.c_void_p),
                                              None if m is None else m.ctypes.data_as(ctypes.c_void_p),
                                              out_s.ctypes.data_as(ctypes.c_void_p), out_i.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.byref(scanned)))
        return out_s, out_i, int(scanned.value)

    def search_delta_device(self, flat: FlatIndex, d_queries_ptr: int, nq: int, k: int, nprobe: int, d_out_scores_ptr: int,
                            d_out_ids_ptr: int, d_q_filter_ptr: int = 0, d_q_filter_mask_ptr: int = 0) -> None:
        N.check("rass_ivf_search_delta_device",
                self._L.rass_ivf_search_delta_device(self._h, flat._h, ctypes.c_void_p(d_queries_ptr), int(nq), int(k),
                                                     int(nprobe), ctypes.c_void_p(d_q_filter_ptr or 0),
                                                     ctypes.c_void_p(d_q_filter_mask_ptr or 0),
                                                     ctypes.c_void_p(d_out_scores_ptr), ctypes.c_void_p(d_out_ids_ptr)))

    @property
    def dtype(self) -> str:
        return {v: k for k, v in SLAB_DTYPES.items()}[int(self._L.rass_ivf_dtype(self._h))]

    @property
    def max_k(self) -> int:
        """The largest k a probe serves: 32, or 16 over an int8 slab (its 32 candidates per query are re-ranked)."""
        return 16 if self.dtype == "int8" else 32

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


# ------------------------------------------------------------------------------------ IVF behind the boundary
class IvfPolicy:
    """When an index gets an IVF and when that IVF is rebuilt (``RASS_IVF_*``, config.py).

    The reference's index is approximate (HNSW, app/main.py:563-572) AND takes inserts at any time (bulk per 64 docs,
    app/main.py:1253-1282).  Here an index below ``min_rows`` stays flat (an exact scan of 262 144 rows is 0.16 ms);
    above it an IVF-``nlist`` is built over the rows it holds; rows appended afterwards form a flat DELTA that every
    search scans exactly next to the probe (``rass_ivf_search_delta``), and once the delta exceeds ``rebuild_fraction``
    of the covered rows the IVF is rebuilt over everything (2.4-3.7 s per 12.5 M rows, inside the ``add`` that crossed
    the threshold — an ingest-path cost; searches keep being served from the old IVF + delta until the swap)."""

    def __init__(self, nlist: int = 0, nprobe: int = 8, min_rows: int = 262144, rebuild_fraction: float = 0.25,
                 dtype: str = "f32", train_rows: int = 0, iters: int = 10, seed: int = 0):
        self.nlist, self.nprobe, self.min_rows = int(nlist), max(1, int(nprobe)), int(min_rows)
        self.rebuild_fraction, self.dtype = float(rebuild_fraction), dtype
        self.train_rows, self.iters, self.seed = int(train_rows), int(iters), int(seed)

    @classmethod
    def from_config(cls) -> "IvfPolicy":
        from . import config
        return cls(config.RASS_IVF_NLIST, config.RASS_IVF_NPROBE, config.RASS_IVF_MIN_ROWS,
                   config.RASS_IVF_REBUILD_FRACTION, config.RASS_IVF_DTYPE)

    @classmethod
    def manual(cls, nprobe: int = 8) -> "IvfPolicy":
        """No automatic builds (a shard of a multi-GPU index: rank 0 decides for all ranks, ``OP_IVF_BUILD``)."""
        return cls(0, nprobe)

    def due(self, rows: int, covered: int) -> bool:
        if self.nlist <= 0 or rows < max(self.min_rows, self.nlist):
            return False
        if covered <= 0:
            return True
        return rows - covered > self.rebuild_fraction * covered


class IvfBackedIndex(FlatIndex):
    """A ``FlatIndex`` (same handle, same surface: what ``IndexState.index`` / ``HipServingShard`` talk to) whose
    searches go through an IVF over its first ``covered`` rows + an exact scan of the rows appended since, merged by
    the engine's merge kernel; tombstones are applied to both.  The flat index stays authoritative: k > 32, continuation
    passes, ``get_row``, rebuilds and every search before the first build are served from it exactly."""

    def __init__(self, flat: FlatIndex, policy: Optional[IvfPolicy] = None):
        super().__init__(flat.engine, flat.name, flat._h)
        import threading
        self.policy = policy or IvfPolicy.from_config()
        self.ivf: Optional[IvfIndex] = None
        self.builds = 0                      # bumped by every swap (part of ``epoch``)
        self._ivf_lock = threading.RLock()   # deletes and the build's final phase exclude each other

    # ---- bookkeeping
    @property
    def covered(self) -> int:
        ivf = self.ivf
        return ivf.covered_rows if ivf is not None else 0

    @property
    def epoch(self) -> Tuple[int, int, int]:
        rows = self.rows
        return rows, rows - self.count, self.builds

    @property
    def multi_tiles(self) -> int:
        """An index with an IVF answers through its own search (it cannot join a cross-index work-list launch)."""
        return 0 if self.ivf is not None else super().multi_tiles

    # ---- build / rebuild
    def build_ivf(self, nlist: Optional[int] = None, group: Optional[dist.ProcessGroup] = None,
                  centroids: Optional[torch.Tensor] = None, dtype: Optional[str] = None) -> Optional[IvfIndex]:
        """Train (all ranks of ``group`` together: shared centroids), assign, and build the IVF over the first
        ``rows // 32 * 32`` rows (the scan's tile: the delta must start on a tile); the remaining <= 31 rows and
        everything appended later are the delta.  Swaps the new IVF in and frees the old one."""
        p = self.policy
        nlist = int(nlist or p.nlist)
        if nlist <= 0:
            raise ValueError("build_ivf needs nlist > 0 (RASS_IVF_NLIST)")
        if centroids is None:
            centroids = train_centroids(self, nlist, p.train_rows, p.iters, p.seed, group)
        rows = self.rows
        assign = assign_rows(self, centroids)
        with self._ivf_lock:
            new = IvfIndex.build(self, nlist=nlist, centroids=centroids, dtype=dtype or p.dtype, assign=assign,
                                 n_rows=min(rows, len(assign)) // 32 * 32)
            old, self.ivf = self.ivf, new
            self.builds += 1
        if old is not None:
            old.close()
        return new

    def maybe_rebuild(self) -> bool:
        if self.policy.due(self.rows, self.covered):
            self.build_ivf()
            return True
        return False

    def drop_ivf(self) -> None:
        with self._ivf_lock:
            old, self.ivf = self.ivf, None
            self.builds += 1
        if old is not None:
            old.close()

    # ---- write path
    def add(self, vecs, tags=None, normalize: bool = True, first_global_id: int = -1) -> int:
        first = super().add(vecs, tags=tags, normalize=normalize, first_global_id=first_global_id)
        self.maybe_rebuild()
        return first

    def add_device(self, d_vecs_ptr: int, n: int, d_tags_ptr: int = 0, normalize: bool = True) -> int:
        first = super().add_device(d_vecs_ptr, n, d_tags_ptr, normalize)
        self.maybe_rebuild()
        return first

    def delete(self, row: int) -> None:
        with self._ivf_lock:
            super().delete(row)
            if self.ivf is not None:
                self.ivf.delete(row)

    # ---- persistence: `<path>` = the flat index (authoritative), `<path minus .tmp>.ivf` = the IVF's device state
    @staticmethod
    def _ivf_path(path: str) -> str:
        return (path[:-4] if path.endswith(".tmp") else path) + ".ivf"

    def save(self, path: str) -> None:
        with self._ivf_lock:
            if self.ivf is not None:
                self.ivf.save(self._ivf_path(path))
            super().save(path)

    def saved_files(self, path: str):
        """Files of a saved generation besides ``path`` itself (``IndexState.save`` removes the previous one's)."""
        return [self._ivf_path(path)]

    @classmethod
    def load(cls, engine, name: str, path: str, policy: Optional[IvfPolicy] = None) -> "IvfBackedIndex":
        import os
        idx = cls(engine.load_index(name, path), policy)
        f = cls._ivf_path(path)
        if os.path.exists(f):
            try:
                ivf = IvfIndex.load(engine, f)
                if ivf.covered_rows <= idx.rows and (ivf.covered_rows % 32 == 0 or ivf.covered_rows == idx.rows):
                    idx.ivf = ivf
                else:
                    ivf.close()
            except Exception:       # a missing / torn IVF file costs a rebuild, never the index
                idx.ivf = None
        return idx

    # ---- read path
    def _use_ivf(self, k: int) -> Optional[IvfIndex]:
        ivf = self.ivf
        return ivf if (ivf is not None and 1 <= int(k) <= ivf.max_k) else None

    def search(self, queries: np.ndarray, k: int, q_filter: Optional[np.ndarray] = None,
               q_filter_mask: Optional[np.ndarray] = None, exact: bool = False, nprobe: Optional[int] = None
               ) -> Tuple[np.ndarray, np.ndarray]:
        ivf = None if exact else self._use_ivf(k)
        if ivf is None:
            return super().search(queries, k, q_filter, q_filter_mask)
        s, i, _ = ivf.search_delta(self, queries, int(k), int(nprobe or self.policy.nprobe), q_filter, q_filter_mask)
        return s, i

    def search_device(self, d_queries_ptr: int, nq: int, k: int, d_out_scores_ptr: int, d_out_ids_ptr: int,
                      id_base: int = 0, d_q_filter_ptr: int = 0, d_q_filter_mask_ptr: int = 0, exact: bool = False,
                      nprobe: Optional[int] = None) -> None:
        ivf = None if (exact or id_base) else self._use_ivf(k)
        if ivf is None:
            return super().search_device(d_queries_ptr, nq, k, d_out_scores_ptr, d_out_ids_ptr, id_base, d_q_filter_ptr,
                                         d_q_filter_mask_ptr)
        ivf.search_delta_device(self, d_queries_ptr, nq, k, int(nprobe or self.policy.nprobe), d_out_scores_ptr,
                                d_out_ids_ptr, d_q_filter_ptr, d_q_filter_mask_ptr)


def open_backed_index(engine, name: str, policy: Optional[IvfPolicy] = None) -> IvfBackedIndex:
    """``docstore.REGISTRY``'s index factory when ``RASS_IVF_NLIST`` > 0."""
    return IvfBackedIndex(engine.open_index(name), policy)
