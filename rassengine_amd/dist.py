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

    def search_local_packed_into(self, queries: torch.Tensor, k: int, rec: torch.Tensor) -> None:
        """Same, into a caller-owned record (a slice of a batch's record buffer)."""
        nq = queries.shape[0]
        ids_off, size = self.record_bytes(nq, k)
        assert rec.numel() >= size and rec.data_ptr() % 8 == 0
        self.index.search_device(queries.data_ptr(), nq, k, rec.data_ptr(), rec.data_ptr() + ids_off,
                                 id_base=self.id_base)

    def merge_packed_group(self, gathered: torch.Tensor, world: int, group: int, groups: int, nq: int, k: int,
                           out_s: torch.Tensor, out_i: torch.Tensor) -> None:
        """Merge launch group ``group`` of a gathered [world][groups][record] buffer into out_s / out_i."""
        import ctypes
        from . import _native as N
        ids_off, size = self.record_bytes(nq, k)
        base = gathered.data_ptr() + group * size
        N.check("rass_topk_merge_strided",
                N.lib().rass_topk_merge_strided(ctypes.c_void_p(base), ctypes.c_void_p(base + ids_off), groups * size // 4,
                                                groups * size // 8, world, nq, k, ctypes.c_void_p(out_s.data_ptr()),
                                                ctypes.c_void_p(out_i.data_ptr()),
                                                ctypes.c_void_p(int(torch.cuda.current_stream(self.device).cuda_stream))))

    def search_local_packed_batch_into(self, queries: torch.Tensor, k: int, group_size: int, recs: torch.Tensor) -> None:
        """All launch groups of a batch in ONE engine call (``rass_index_search_device_batch``): group g's packed
        record lands at recs[g * record_size:].  Needs group_size = 32, the engine's launch group."""
        n = queries.shape[0]
        ids_off, size = self.record_bytes(group_size, k)
        assert group_size == 32 and n % group_size == 0 and recs.numel() >= (n // group_size) * size
        assert recs.data_ptr() % 8 == 0 and queries.is_contiguous()
        self.index.search_device_batch(queries.data_ptr(), n, k, recs.data_ptr(), recs.data_ptr() + ids_off,
                                       id_base=self.id_base, out_scores_group_stride=size // 4,
                                       out_ids_group_stride=size // 8)

    def merge_packed_batch(self, gathered: torch.Tensor, world: int, groups: int, group_size: int, k: int,
                           out_s: torch.Tensor, out_i: torch.Tensor) -> None:
        """ONE merge launch over a gathered [world][groups][record] buffer -> out_s / out_i [groups * group_size, k]."""
        import ctypes
        from . import _native as N
        ids_off, size = self.record_bytes(group_size, k)
        base = gathered.data_ptr()
        N.check("rass_topk_merge_strided_batch",
                N.lib().rass_topk_merge_strided_batch(
                    ctypes.c_void_p(base), ctypes.c_void_p(base + ids_off), groups * size // 4, groups * size // 8, world,
                    groups * group_size, group_size, size // 4, size // 8, k, ctypes.c_void_p(out_s.data_ptr()),
                    ctypes.c_void_p(out_i.data_ptr()),
                    ctypes.c_void_p(int(torch.cuda.current_stream(self.device).cuda_stream))))

    def search_local_batch(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Single shard: a whole batch [n, dim] -> (scores [n, k], ids [n, k]) in one engine call."""
        n = queries.shape[0]
        out_s = torch.empty((n, k), dtype=torch.float32, device=self.device)
        out_i = torch.empty((n, k), dtype=torch.int64, device=self.device)
        self.index.search_device_batch(queries.data_ptr(), n, k, out_s.data_ptr(), out_i.data_ptr(), id_base=self.id_base)
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


def search_batch(self: "ShardedSearch", queries: torch.Tensor, k: int, group_size: int = 32, src: int = 0
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """A whole batch of queries [n, dim] (n a multiple of ``group_size`` <= 32, the kernel's queries per launch)
    with TWO collectives however many launch groups it takes: one broadcast of the batch, the groups' local
    scans into one record buffer, ONE all-gather of [groups][record], one strided merge per group.  Returns
    (scores [n, k], global ids [n, k]) on every rank.  (Per-group ``search`` costs a broadcast + an all-gather
    per 32 queries: 64 collectives for a 1 024-query batch.)"""
    n = queries.shape[0]
    assert n % group_size == 0 and getattr(self.shard, "packed", False)
    groups = n // group_size
    if self.world > 1:
        dist.broadcast(queries, src=src, group=self.group)
    elif group_size == 32 and hasattr(self.shard, "search_local_batch"):
        return self.shard.search_local_batch(queries, k)   # one shard: nothing to exchange or merge again
    dev = queries.device
    _, size = HipShard.record_bytes(group_size, k)
    recs = torch.empty((groups * size,), dtype=torch.uint8, device=dev)
    one_call = group_size == 32 and hasattr(self.shard, "search_local_packed_batch_into")
    if one_call:    # every group's scan in one engine call: one normalise + one merge launch for the batch
        self.shard.search_local_packed_batch_into(queries, k, group_size, recs)
    else:
        for g in range(groups):
            self.shard.search_local_packed_into(queries[g * group_size:(g + 1) * group_size], k, recs[g * size:(g + 1) * size])
    if self.world > 1:
        gathered = torch.empty((self.world * recs.numel(),), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, recs, group=self.group)
    else:
        gathered = recs
    out_s = torch.empty((n, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((n, k), dtype=torch.int64, device=dev)
    if one_call:
        self.shard.merge_packed_batch(gathered, self.world, groups, group_size, k, out_s, out_i)
    else:
        for g in range(groups):
            self.shard.merge_packed_group(gathered, self.world, g, groups, group_size, k,
                                          out_s[g * group_size:(g + 1) * group_size], out_i[g * group_size:(g + 1) * group_size])
    return out_s, out_i


ShardedSearch.search_batch = search_batch


class PeerMergeSearch:
    """``ShardedSearch`` with the all-gather replaced by peer stores (``csrc/peer.hip``).

    Rank 0 owns a buffer ``[G flags, 64 B apart][2 parities][G record slots]`` in its HBM; the other ranks map
    it through a HIP IPC handle (exchanged once, at construction).  A step: broadcast(queries) -> every rank scans
    its shard into a packed record and STORES it into its slot of parity ``step & 1`` (over xGMI between GPUs),
    then releases its flag = step -> rank 0 enqueues the bounded flag wait and the strided merge.  Only rank 0
    gets the result (the coordinator; ``search`` returns ``None`` elsewhere).  Results are identical to
    ``ShardedSearch`` (same records, same merge kernel).  Validated with 2 ranks sharing one GPU
    (tests/test_gpu_dist.py); between GPUs it has not run yet (no multi-GPU box in this build)."""

    FLAG_STRIDE = 64
    MAX_SPINS = 1 << 22          # x s_sleep(8): gives up after ~1 s instead of hanging the GPU

    def __init__(self, shard: "HipShard", nq_max: int = 32, k_max: int = 32, group: Optional[dist.ProcessGroup] = None):
        import ctypes
        from . import _native as N
        self.shard = shard
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = shard.device
        _, self.slot_bytes = HipShard.record_bytes(nq_max, k_max)
        self.slot_bytes = (self.slot_bytes + 255) // 256 * 256
        self.flags_bytes = (self.world * self.FLAG_STRIDE + 255) // 256 * 256
        total = self.flags_bytes + 2 * self.world * self.slot_bytes
        L = N.lib()
        ptr = ctypes.c_void_p()
        handle = ctypes.create_string_buffer(64)
        self._opened = False
        if self.rank == 0:
            N.check("rass_peer_buffer_create", L.rass_peer_buffer_create(self.device.index, total, ctypes.byref(ptr), handle))
        box = [bytes(handle.raw) if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        if self.rank != 0:
            N.check("rass_peer_buffer_open", L.rass_peer_buffer_open(self.device.index, box[0], ctypes.byref(ptr)))
            self._opened = True
        self._base = int(ptr.value)
        self._L = L
        self._N = N
        self.step = 0
        self._status = torch.zeros((1,), dtype=torch.int32, device=self.device) if self.rank == 0 else None

    def close(self) -> None:
        if getattr(self, "_base", 0):
            torch.cuda.synchronize(self.device)
            if self.world > 1:
                dist.barrier(group=self.group)       # nobody unmaps / frees while a peer may still store
            import ctypes
            if self._opened:
                self._N.check("rass_peer_buffer_close", self._L.rass_peer_buffer_close(ctypes.c_void_p(self._base), 1))
            if self.world > 1:
                dist.barrier(group=self.group)       # rank 0 frees after every peer has unmapped
            if not self._opened:
                self._N.check("rass_peer_buffer_close", self._L.rass_peer_buffer_close(ctypes.c_void_p(self._base), 0))
            self._base = 0

    def search(self, queries: torch.Tensor, k: int, src: int = 0):
        import ctypes
        if self.world > 1:
            dist.broadcast(queries, src=src, group=self.group)
        nq = queries.shape[0]
        self.step += 1
        seq = self.step
        rec = self.shard.search_local_packed(queries, k)
        ids_off, size = HipShard.record_bytes(nq, k)
        size16 = (size + 15) // 16 * 16
        if rec.numel() < size16:                    # the copy kernel moves whole 16-byte pieces
            rec = torch.cat([rec, torch.zeros((size16 - rec.numel(),), dtype=torch.uint8, device=rec.device)])
        slots = self._base + self.flags_bytes + (seq & 1) * self.world * self.slot_bytes
        stream = ctypes.c_void_p(int(torch.cuda.current_stream(self.device).cuda_stream))
        self._N.check("rass_peer_post", self._L.rass_peer_post(
            ctypes.c_void_p(rec.data_ptr()), size16, ctypes.c_void_p(slots + self.rank * self.slot_bytes),
            ctypes.c_void_p(self._base + self.rank * self.FLAG_STRIDE), seq, stream))
        if self.rank != 0:
            return None
        self._N.check("rass_peer_wait", self._L.rass_peer_wait(
            ctypes.c_void_p(self._base), self.world, self.FLAG_STRIDE, seq, ctypes.c_void_p(self._status.data_ptr()),
            self.MAX_SPINS, stream))
        out_s = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        self._N.check("rass_topk_merge_strided", self._L.rass_topk_merge_strided(
            ctypes.c_void_p(slots), ctypes.c_void_p(slots + ids_off), self.slot_bytes // 4, self.slot_bytes // 8,
            self.world, nq, k, ctypes.c_void_p(out_s.data_ptr()), ctypes.c_void_p(out_i.data_ptr()), stream))
        return out_s, out_i

    def check(self) -> None:
        """Rank 0: raise if a wait gave up (a peer never posted)."""
        if self.rank == 0:
            st = int(self._status.item())
            if st:
                raise RuntimeError(f"peer-store exchange: rank {st - 1} never posted its record")
