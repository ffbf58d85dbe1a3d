"""Row-sharded exact search across the GPUs of one node.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI).  The corpus
is partitioned into contiguous row blocks, rank r owning global rows
[r*N/G, (r+1)*N/G); every rank scans its block for the same query batch; the only exchange
is ONE all-gather of the per-shard top-k (``nq*k*12`` bytes per rank: latency-, not
bandwidth-bound) after which every rank merges G*k -> k with the scan's total order, so all
ranks hold the identical result and it equals the single-GPU result bit for bit.  This is
the reference's OpenSearch shard -> coordinator merge (``SHARD_COUNT``, app/main.py:89, 357)
done over xGMI instead of HTTP.

Ingest is pure data-parallel (no collective): each rank appends to its own shard.
"""
from __future__ import annotations

from typing import List, Optional, Protocol, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous row ranges [lo, hi) per rank; row id = global ordinal."""
    return [(r * n_total // world, (r + 1) * n_total // world) for r in range(world)]


class LocalShard(Protocol):
    """What the sharded search needs from one rank's shard."""

    device: torch.device

    def search_local(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(scores f32 [nq,k], GLOBAL ids i64 [nq,k]) for this shard, best first."""

    def merge(self, list_scores: torch.Tensor, list_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """[G,nq,k] candidate lists -> [nq,k] under (score desc, id asc)."""


class HipShard:
    """A rank's shard backed by a ``FlatIndex`` in HBM; search and merge are HIP kernels
    enqueued on torch's current stream (which the engine is switched to), so the RCCL
    collectives that follow are ordered after them without a host sync.

    The local result is written straight into ONE packed record (nq*k f32 scores, then nq*k
    i64 ids, 8-byte aligned), so the exchange is a single all-gather and the merge kernel
    reads the gathered records in place (``rass_topk_merge_strided``)."""

    packed = True

    def __init__(self, index, id_base: int):
        from . import ops  # noqa: F401  (fails loudly without the HIP library)
        self.index = index
        self.id_base = int(id_base)
        self.device = torch.device("cuda", index.engine.device)
        index.engine.set_stream(int(torch.cuda.current_stream(self.device).cuda_stream))

    @staticmethod
    def record_bytes(nq: int, k: int) -> Tuple[int, int]:
        """(offset of the ids inside a record, record size), both multiples of 8."""
        ids_off = (nq * k * 4 + 7) // 8 * 8
        return ids_off, ids_off + nq * k * 8

    def search_local_packed(self, queries: torch.Tensor, k: int) -> torch.Tensor:
        nq = queries.shape[0]
        ids_off, size = self.record_bytes(nq, k)
        rec = torch.empty((size,), dtype=torch.uint8, device=self.device)
        self.index.search_device(queries.data_ptr(), nq, k, rec.data_ptr(), rec.data_ptr() + ids_off,
                                 id_base=self.id_base)
        return rec

    def merge_packed(self, gathered: torch.Tensor, world: int, nq: int, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        import ctypes
        from . import _native as N
        ids_off, size = self.record_bytes(nq, k)
        out_s = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        base = gathered.data_ptr()
        N.check("rass_topk_merge_strided",
                N.lib().rass_topk_merge_strided(ctypes.c_void_p(base), ctypes.c_void_p(base + ids_off), size // 4,
                                                size // 8, world, nq, k, ctypes.c_void_p(out_s.data_ptr()),
                                                ctypes.c_void_p(out_i.data_ptr()),
                                                ctypes.c_void_p(int(torch.cuda.current_stream(self.device).cuda_stream))))
        return out_s, out_i

    def search_local(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        nq = queries.shape[0]
        out_s = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        self.index.search_device(queries.data_ptr(), nq, k, out_s.data_ptr(), out_i.data_ptr(), id_base=self.id_base)
        return out_s, out_i

    def merge(self, list_scores: torch.Tensor, list_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        from . import ops
        return ops.topk_merge(list_scores, list_ids)


class ShardedSearch:
    """broadcast queries -> local scan -> all-gather(top-k) -> merge, on every rank."""

    def __init__(self, shard: LocalShard, group: Optional[dist.ProcessGroup] = None):
        self.shard = shard
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def search(self, queries: torch.Tensor, k: int, src: int = 0, broadcast: bool = True
               ) -> Tuple[torch.Tensor, torch.Tensor]:
        """``queries`` [nq, dim] must be allocated on every rank; rank ``src`` holds the data.
        Returns (scores [nq,k], global ids [nq,k]), identical on every rank."""
        if self.world > 1 and broadcast:
            dist.broadcast(queries, src=src, group=self.group)
        nq = queries.shape[0]
        if self.world > 1 and getattr(self.shard, "packed", False):
            # one collective: every rank contributes one packed (scores | ids) record
            rec = self.shard.search_local_packed(queries, k)
            gathered = torch.empty((self.world * rec.numel(),), dtype=torch.uint8, device=rec.device)
            dist.all_gather_into_tensor(gathered, rec, group=self.group)
            return self.shard.merge_packed(gathered, self.world, nq, k)
        loc_s, loc_i = self.shard.search_local(queries, k)
        if self.world == 1:
            return loc_s, loc_i
        gath_s = torch.empty((self.world, nq, k), dtype=loc_s.dtype, device=loc_s.device)
        gath_i = torch.empty((self.world, nq, k), dtype=loc_i.dtype, device=loc_i.device)
        # output = concatenation along dim 0 of the per-rank [nq, k] blocks (rank-major)
        dist.all_gather_into_tensor(gath_s.view(self.world * nq, k), loc_s.contiguous(), group=self.group)
        dist.all_gather_into_tensor(gath_i.view(self.world * nq, k), loc_i.contiguous(), group=self.group)
        return self.shard.merge(gath_s, gath_i)
