"""The multi-GPU index BEHIND the reference boundary.

``dist.ShardedSearch`` is the collective step (broadcast -> local scan -> one all-gather -> merge); this
module puts it behind ``OpenSearchIndexer``: one process per GPU (``torchrun --nproc-per-node G``), rank 0 runs
the FastAPI process and serves ``HipIndexer`` over a ``ShardedIndex`` front, ranks 1..G-1 sit in
``worker_loop`` and follow rank 0's command stream.  The reference's analogue is OpenSearch's
``number_of_shards = SHARD_COUNT`` (app/main.py:89, 357) with the coordinator merging per-shard top-k; here the
shards are GPUs, the coordinator is rank 0 and the exchange is ONE RCCL all-gather of packed per-shard top-k.

Command stream: every operation starts with ONE broadcast (src 0) of a 16 x int64 header over a HOST-side gloo
group (``ShardServer.ctl``) — idle workers block in a socket read: no RCCL kernel spins on their GPUs and no NCCL
watchdog (10-minute default) can time the wait out; the group is created with a timeout of years — followed, when
the operation has one, by a broadcast of its payload (queries / filters / name) over the data group (RCCL); every
rank then executes the same operation on its shard, so the collectives line up by construction:

    OPEN    every rank opens its local shard of the named index
    ADD     rows are dealt to the ranks ROUND-ROBIN BY BATCH (batch b -> rank b mod G); the batch travels
            point-to-point to its owner only; its rows get consecutive GLOBAL row ids (insertion ordinals of the
            whole index = the single-GPU ids) through rass_index_add_ex, so a shard reports global ids itself
    SEARCH  local masked scan -> packed (scores | ids) record -> one all-gather -> merge on rank 0
    DELETE  the owner (global id -> (rank, ordinal) through the extent table) tombstones the row
    COUNT   all-reduce of the shards' live row counts
    GETROW  the owner sends the stored row to rank 0
    ENCODE  data-parallel ingest: rank 0 tokenises, the round's token ids are broadcast, every rank encodes the
            batches dealt to it with its own encoder straight into its shard (no embedding crosses ranks)
    SAVE    every rank writes its shard file, an all-reduce tells rank 0 that all are durable, rank 0 writes the
            manifest (world size, run table, tombstones); LOAD is the inverse and rebuilds the ranks' extent tables
    IVF_BUILD every rank trains the SAME centroids together (k-means sums / counts all-reduced over the data group),
            assigns its own rows and builds an IVF over its shard; later SEARCHes probe it (+ an exact scan of the rows
            the shard took since: the flat delta) unless they carry the EXACT flag (k > 32 passes)
    SHUTDOWN workers leave the loop; every rank then meets in a barrier (a clean collective exit: the workers
            are ordinary processes that return, nothing is re-exec'ed or killed)

Failures: the LOCAL part of every operation runs under try / except on every rank, the operation's data collectives
are carried out regardless (with stand-in data), and the operation ends with a verdict every rank takes part in (an
all-gather of one status word; SEARCH carries it in the tail of the record it gathers anyway): either no rank raises,
or EVERY rank raises ``CollectiveFailure`` at the same point of the stream — a worker logs it and keeps following
rank 0, rank 0 reports it to its caller.  Rank 0 commits its bookkeeping (rows, runs, owners) only after the verdict;
global ids an operation may already have consumed on some rank are never reused (they become holes).

Because a shard's ids ascend with its append order and ties are broken by (score desc, id asc) both inside a
shard and in the merge, the sharded result equals the single-index result bit for bit (tests/test_serving_gloo.py;
tests/test_gpu_dist.py for the HIP shards).
"""
from __future__ import annotations

import bisect
import datetime
import json
import logging
import os
import threading
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

logger = logging.getLogger("rassengine_amd.serving")

OP_SHUTDOWN, OP_OPEN, OP_SEARCH, OP_ADD, OP_DELETE, OP_COUNT, OP_GETROW, OP_DROP, OP_ENCODE, OP_SAVE, OP_LOAD, OP_IVF_BUILD = range(12)
SEARCH_FILTER, SEARCH_MASK, SEARCH_AFTER, SEARCH_EXACT = 1, 2, 4, 8     # flag bits of OP_SEARCH
ENC_BATCH_SEQS = 256          # sequences per encoder batch = the unit dealt to a rank (BASELINE configs[2]: batch 256)
HDR_WORDS = 16
MAX_Q = 32
MAX_K = 32
NAME_BYTES = 1024
ADD_CHUNK_ROWS = 8192


CTL_TIMEOUT = datetime.timedelta(days=3650)   # an idle service is not an error


class CollectiveFailure(RuntimeError):
    """An operation failed on some rank; EVERY rank raises this at the same point of the command stream (the verdict
    comes out of a collective), so the workers can log it and keep following rank 0 while rank 0 reports it.
    ``failed_ranks``: who failed."""

    def __init__(self, msg: str, failed_ranks=()):
        super().__init__(msg)
        self.failed_ranks = tuple(int(r) for r in failed_ranks)


class Extents:
    """Which global row ids a rank holds: sorted (gid_base, ordinal_base, n) runs."""

    def __init__(self):
        self.gid: List[int] = []
        self.ordinal: List[int] = []
        self.n: List[int] = []

    def append(self, gid_base: int, ordinal_base: int, n: int) -> None:
        if self.gid and self.gid[-1] + self.n[-1] == gid_base and self.ordinal[-1] + self.n[-1] == ordinal_base:
            self.n[-1] += n
            return
        self.gid.append(gid_base)
        self.ordinal.append(ordinal_base)
        self.n.append(n)

    def ordinal_upto(self, gid: int) -> int:
        """The largest local ordinal whose global id is <= ``gid`` (-1: none).  Ids ascend with ordinals, so "rows with
        a global id > gid" are exactly "rows with an ordinal > ordinal_upto(gid)" (continuation bound of a k > 32 pass)."""
        e = bisect.bisect_right(self.gid, gid) - 1
        if e < 0:
            return -1
        return self.ordinal[e] + min(gid - self.gid[e], self.n[e] - 1)

    def ordinal_of(self, gid: int) -> Optional[int]:
        e = bisect.bisect_right(self.gid, gid) - 1
        if e < 0 or gid >= self.gid[e] + self.n[e]:
            return None
        return self.ordinal[e] + (gid - self.gid[e])


class HipServingShard:
    """One rank's shard of a served index: a ``FlatIndex`` in HBM (``LocalServingShard`` surface)."""

    def __init__(self, index):
        from . import ops  # noqa: F401  (fails loudly without the HIP library)
        self.index = index
        self.device = torch.device("cuda", index.engine.device)
        index.engine.set_stream(int(torch.cuda.current_stream(self.device).cuda_stream))

    dim = property(lambda self: self.index.dim)
    count = property(lambda self: self.index.count)
    rows = property(lambda self: self.index.rows)

    def add(self, vecs: torch.Tensor, tags: torch.Tensor, normalize: bool, first_global_id: int) -> int:
        import ctypes
        from . import _native as N
        first = ctypes.c_int64(-1)
        torch.cuda.current_stream(self.device).synchronize()     # the batch arrived through a collective
        N.check("rass_index_add_ex", N.lib().rass_index_add_ex(
            self.index._h, ctypes.c_void_p(vecs.data_ptr()), ctypes.c_void_p(tags.data_ptr()), int(vecs.shape[0]),
            1 if normalize else 0, int(first_global_id), 1, ctypes.byref(first)))
        return int(first.value)

    def delete(self, ordinal: int) -> None:
        self.index.delete(ordinal)

    def save(self, path: str) -> None:
        self.index.save(path)

    def get_row(self, ordinal: int) -> torch.Tensor:
        return torch.from_numpy(self.index.get_row(ordinal)).to(self.device)

    @staticmethod
    def record_bytes(nq: int, k: int) -> Tuple[int, int]:
        ids_off = (nq * k * 4 + 7) // 8 * 8
        return ids_off, ids_off + nq * k * 8

    strided_records = True     # search_packed(out=...) writes in place, merge_packed(stride_bytes=...) reads records with tails

    def build_ivf(self, nlist: int, nprobe: int, dtype: str = "f32", group=None) -> None:
        """This rank's part of OP_IVF_BUILD: shared centroids (trained by all ranks of ``group`` together), an IVF over
        this shard's rows.  The shard's index must be an ``ivf.IvfBackedIndex`` (``hip_shard_factory`` makes one)."""
        self.index.policy.nprobe = max(1, int(nprobe))
        self.index.build_ivf(nlist=int(nlist), group=group, dtype=dtype)

    def search_packed(self, queries: torch.Tensor, k: int, filt: Optional[torch.Tensor], mask: Optional[torch.Tensor],
                      after: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, out: Optional[torch.Tensor] = None,
                      exact: bool = False, nprobe: int = 0) -> torch.Tensor:
        """``after`` = (scores f32 [nq], LOCAL row ordinals i64 [nq]): rank only the rows strictly behind them.
        ``out``: a caller-owned buffer of at least the record size (8-byte aligned) to write the record into.
        ``exact``: scan the flat shard even when it has an IVF (the passes of a k > 32 search); ``nprobe``: lists per
        query of an IVF probe (0 = the shard's own default) — a query-time parameter, so it travels with the command."""
        nq = queries.shape[0]
        ids_off, size = self.record_bytes(nq, k)
        rec = torch.empty((size,), dtype=torch.uint8, device=self.device) if out is None else out
        if after is not None:
            a_s, a_r = after[0].to(self.device).contiguous(), after[1].to(self.device).contiguous()
            self.index.search_device_after(queries.data_ptr(), nq, k, a_s.data_ptr(), a_r.data_ptr(), rec.data_ptr(),
                                           rec.data_ptr() + ids_off,
                                           d_q_filter_ptr=filt.data_ptr() if filt is not None else 0,
                                           d_q_filter_mask_ptr=mask.data_ptr() if mask is not None else 0)
            self._keep = (a_s, a_r)      # alive until the stream has consumed them
            return rec
        kw = {}
        if hasattr(self.index, "build_ivf"):
            kw = {"exact": bool(exact), "nprobe": int(nprobe) or None}
        self.index.search_device(queries.data_ptr(), nq, k, rec.data_ptr(), rec.data_ptr() + ids_off, id_base=0,
                                 d_q_filter_ptr=filt.data_ptr() if filt is not None else 0,
                                 d_q_filter_mask_ptr=mask.data_ptr() if mask is not None else 0, **kw)
        return rec

    def merge_packed(self, gathered: torch.Tensor, world: int, nq: int, k: int, stride_bytes: int = 0
                     ) -> Tuple[np.ndarray, np.ndarray]:
        import ctypes
        from . import _native as N
        ids_off, size = self.record_bytes(nq, k)
        stride = stride_bytes or size          # records may carry a tail (the status word of ShardServer's SEARCH)
        out_s = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        base = gathered.data_ptr()
        N.check("rass_topk_merge_strided",
                N.lib().rass_topk_merge_strided(ctypes.c_void_p(base), ctypes.c_void_p(base + ids_off), stride // 4,
                                                stride // 8, world, nq, k, ctypes.c_void_p(out_s.data_ptr()),
                                                ctypes.c_void_p(out_i.data_ptr()),
                                                ctypes.c_void_p(int(torch.cuda.current_stream(self.device).cuda_stream))))
        return out_s.cpu().numpy(), out_i.cpu().numpy()


class ShardServer:
    """Runs on EVERY rank: owns the rank's shards and executes the command stream."""

    def __init__(self, shard_factory: Callable[[str], object], dim: int, device: torch.device,
                 group: Optional[dist.ProcessGroup] = None, encoder_factory: Optional[Callable[[], object]] = None,
                 shard_loader: Optional[Callable[[str, str], object]] = None,
                 ctl_group: Optional[dist.ProcessGroup] = None):
        self.factory = shard_factory
        self.loader = shard_loader               # (name, file) -> shard, for OP_LOAD
        self.encoder_factory = encoder_factory   # rank-local sentence encoder (data-parallel ingest), built lazily
        self._encoder = None
        self.dim = dim
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.payload_bytes = max(MAX_Q * dim * 4 + 3 * MAX_Q * 4 + MAX_Q * 8, NAME_BYTES)   # queries|filter|mask|after
        self.cmd = torch.zeros((self.payload_bytes,), dtype=torch.uint8, device=device)      # the payload, on the device
        self.hdr = torch.zeros((HDR_WORDS,), dtype=torch.int64)                              # the header, on the host
        # the control channel: a gloo group of the same ranks (collective creation: every rank constructs its
        # ShardServer at the same point, in serving.start)
        self.ctl = ctl_group if ctl_group is not None else \
            dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None else None,
                           backend="gloo", timeout=CTL_TIMEOUT)
        self.shards: Dict[int, object] = {}
        self.extents: Dict[int, Extents] = {}
        # gloo moves device tensors in its collectives by staging, but its point-to-point ops want host memory
        self._p2p_on_host = dist.get_backend(group) == "gloo" and device.type != "cpu"

    def encoder(self):
        """This rank's encoder: ``tokenize(texts) -> (ids int32 flat, cu int64)`` and ``encode_flat(ids, cu) -> [n, dim]``."""
        if self._encoder is None:
            if self.encoder_factory is None:
                raise RuntimeError("serving.start() was given no encoder_factory: texts cannot be encoded on the ranks")
            self._encoder = self.encoder_factory()
        return self._encoder

    def _bcast(self, t: Optional[torch.Tensor], shape, dtype) -> torch.Tensor:
        """Rank 0's tensor to every rank (the others allocate by the sizes the command header carried)."""
        if self.rank != 0:
            t = torch.empty(shape, dtype=dtype, device=self.device)
        else:
            t = t.to(self.device)
        if self.world > 1:
            dist.broadcast(t, src=0, group=self.group)
        return t

    def _verdict(self, ok: bool, what: str) -> None:
        """Every rank learns whether EVERY rank succeeded, and who did not: all raise ``CollectiveFailure`` or none."""
        t = torch.tensor([0 if ok else 1], dtype=torch.int64, device=self.device)
        if self.world > 1:
            g = torch.empty((self.world,), dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(g, t, group=self.group)
            t = g
        self._raise_if_failed(t.cpu().numpy(), what)

    def _raise_if_failed(self, status: np.ndarray, what: str) -> None:
        failed = [r for r, v in enumerate(np.asarray(status).reshape(-1)) if int(v) != 0]
        if failed:
            raise CollectiveFailure(f"{what} failed on rank(s) {failed}", failed)

    @staticmethod
    def shard_file(base: str, rank: int, world: int) -> str:
        return f"{base}.shard{rank}of{world}"

    def _send(self, t: torch.Tensor, dst: int) -> None:
        dist.send(t.cpu() if self._p2p_on_host else t, dst=dst, group=self.group)

    def _recv(self, shape, dtype, src: int) -> torch.Tensor:
        if self._p2p_on_host:
            h = torch.empty(shape, dtype=dtype)
            dist.recv(h, src=src, group=self.group)
            return h.to(self.device)
        t = torch.empty(shape, dtype=dtype, device=self.device)
        dist.recv(t, src=src, group=self.group)
        return t

    # ---- command transport
    def post(self, header: List[int], payload: Optional[np.ndarray] = None, payload_len: Optional[int] = None) -> np.ndarray:
        """Rank 0: broadcast the header over the host-side control group, then the payload (its first ``payload_len``
        bytes; default: all of what was given) over the data group.  Returns the header."""
        hdr = np.zeros(HDR_WORDS, dtype=np.int64)
        hdr[:len(header)] = header
        nbytes = 0
        if payload is not None:
            raw = np.array(payload, copy=True).view(np.uint8).reshape(-1)     # (np.frombuffer views are read-only)
            nbytes = raw.size if payload_len is None else int(payload_len)
            if nbytes > self.payload_bytes:
                raise ValueError("command payload too large")
        hdr[HDR_WORDS - 1] = nbytes
        self.hdr.copy_(torch.from_numpy(hdr))
        if self.world > 1:
            dist.broadcast(self.hdr, src=0, group=self.ctl)
        if nbytes:
            self.cmd[:nbytes].copy_(torch.from_numpy(raw[:nbytes]))
            if self.world > 1:
                dist.broadcast(self.cmd[:nbytes], src=0, group=self.group)
        return hdr

    def recv(self) -> np.ndarray:
        """Ranks > 0: wait (on the host, in the control group) for rank 0's next command.  Returns the header."""
        dist.broadcast(self.hdr, src=0, group=self.ctl)
        hdr = self.hdr.numpy().copy()
        nbytes = int(hdr[HDR_WORDS - 1])
        if nbytes:
            dist.broadcast(self.cmd[:nbytes], src=0, group=self.group)
        return hdr

    def _payload(self, nbytes: int, offset: int = 0) -> torch.Tensor:
        return self.cmd[offset:offset + nbytes]

    # ---- the operations (identical code on every rank)
    def execute(self, hdr: np.ndarray, vecs: Optional[torch.Tensor] = None, tags: Optional[torch.Tensor] = None,
                enc: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None):
        """One operation, the same code on every rank.  The local part of each operation is fenced by try / except, its
        data collectives always take place, and it ends in a verdict (see the module docstring): a failure on ANY rank
        raises ``CollectiveFailure`` on EVERY rank, after the last collective of the operation."""
        op, code = int(hdr[0]), int(hdr[1])
        ok = True

        def failed(what: str, e: BaseException) -> bool:
            logger.error(f"{what} failed on rank {self.rank}: {e}")
            return False

        if op == OP_OPEN:
            try:
                name = bytes(self._payload(int(hdr[2])).cpu().numpy()).decode("utf-8")
                self.shards[code] = self.factory(name)
                self.extents[code] = Extents()
            except Exception as e:
                ok = failed("open", e)
            try:
                self._verdict(ok, "open")
            except CollectiveFailure:      # an index is open on all ranks or on none
                self.shards.pop(code, None)
                self.extents.pop(code, None)
                raise
            return None
        if op == OP_DROP:
            self.shards.pop(code, None)
            self.extents.pop(code, None)
            return None
        if op == OP_SAVE:
            # every rank writes its shard next to the manifest rank 0 writes afterwards: <base>.shard<r>of<G>,
            # temp name + rename (rass_index_save fsyncs); rank 0 only proceeds when all of them are durable
            try:
                base = bytes(self._payload(int(hdr[2])).cpu().numpy()).decode("utf-8")
                f = self.shard_file(base, self.rank, self.world)
                self.shards[code].save(f + ".tmp")
                os.replace(f + ".tmp", f)
                # this rank's extent table travels with its shard file: an append that failed half-way may have left
                # (tombstoned) rows in the shard that no run of the manifest accounts for, so ordinals are NOT the
                # cumulative sum of the rank's runs in general
                ext = self.extents[code]
                with open(f + ".ext.tmp", "w", encoding="utf-8") as fh:
                    json.dump({"gid": ext.gid, "ordinal": ext.ordinal, "n": ext.n, "rows": int(self.shards[code].rows)}, fh)
                    fh.flush()
                    os.fsync(fh.fileno())
                os.replace(f + ".ext.tmp", f + ".ext")
            except Exception as e:       # reported through the verdict: the ranks must stay in step
                ok = failed("shard save", e)
            self._verdict(ok, "shard save")
            return None
        if op == OP_LOAD:
            plen, nlen, n_runs = int(hdr[2]), int(hdr[3]), int(hdr[4])
            runs = self._bcast(vecs, (max(n_runs, 1), 3), torch.int64).cpu().numpy()[:n_runs]   # (gid, rank, n)
            try:
                base = bytes(self._payload(plen).cpu().numpy()).decode("utf-8")
                name = bytes(self._payload(nlen, plen).cpu().numpy()).decode("utf-8")
                if self.loader is None:
                    raise RuntimeError("serving.start() was given no shard_loader")
                sf = self.shard_file(base, self.rank, self.world)
                shard = self.loader(name, sf)
                ext = Extents()
                mine = [(int(gid), int(n)) for gid, owner, n in runs if int(owner) == self.rank]
                if os.path.exists(sf + ".ext"):
                    # the table this rank saved: ordinals as they really are (rows a failed append left behind included)
                    with open(sf + ".ext", encoding="utf-8") as fh:
                        tab = json.load(fh)
                    for g, o, n in zip(tab["gid"], tab["ordinal"], tab["n"]):
                        ext.append(int(g), int(o), int(n))
                    covered = sorted((g, n) for g, n in zip(ext.gid, ext.n))
                    merged = Extents()
                    for g, n in mine:           # the manifest's runs of this rank, coalesced the same way
                        merged.append(g, merged.ordinal[-1] + merged.n[-1] if merged.gid else 0, n)
                    if int(tab.get("rows", -1)) != int(shard.rows) or sum(n for _, n in covered) != sum(n for _, n in mine) or \
                            any(ext.ordinal_of(g) is None or ext.ordinal_of(g + n - 1) is None for g, n in mine) or \
                            (ext.gid and max(o + n for o, n in zip(ext.ordinal, ext.n)) > int(shard.rows)):
                        raise RuntimeError("the shard's extent table disagrees with the manifest's runs or the shard file")
                else:                           # saved before round 4: ordinals = the cumulative sum of the rank's runs
                    ordinal = 0
                    for gid, n in mine:
                        ext.append(gid, ordinal, n)
                        ordinal += n
                    if ordinal != int(shard.rows):
                        raise RuntimeError(f"shard file holds {int(shard.rows)} rows, the manifest gives this rank {ordinal}")
                self.shards[code], self.extents[code] = shard, ext
            except Exception as e:
                ok = failed("shard load", e)
            try:
                self._verdict(ok, "shard load")
            except CollectiveFailure:
                self.shards.pop(code, None)
                self.extents.pop(code, None)
                raise
            return None
        shard = self.shards.get(code)
        if op == OP_SEARCH:
            nq, k, flags = int(hdr[2]), int(hdr[3]), int(hdr[4])
            ids_off, size = HipServingShard.record_bytes(nq, k)
            # the record every rank contributes carries its status in an 8-byte tail: the all-gather that moves the
            # per-shard top-k IS the verdict (no extra collective on the query path)
            buf = torch.zeros((size + 8,), dtype=torch.uint8, device=self.device)
            try:
                q = self._payload(nq * self.dim * 4).view(torch.float32).view(nq, self.dim)
                off = MAX_Q * self.dim * 4
                filt = self._payload(nq * 4, off).view(torch.int32) if flags & 1 else None
                mask = self._payload(nq * 4, off + MAX_Q * 4).view(torch.int32) if flags & 2 else None
                after = None
                if flags & 4:   # a continuation pass of a k > 32 search: the previous pass's last GLOBAL hit per query
                    a_s = self._payload(nq * 4, off + 2 * MAX_Q * 4).view(torch.float32)
                    a_g = self._payload(nq * 8, off + 3 * MAX_Q * 4).view(torch.int64).cpu().numpy()
                    ext = self.extents[code]
                    a_r = torch.tensor([ext.ordinal_upto(int(g)) for g in a_g], dtype=torch.int64)
                    after = (a_s.clone(), a_r)
                ivf_kw = {"exact": bool(flags & SEARCH_EXACT), "nprobe": int(hdr[5])} if hasattr(shard, "build_ivf") else {}
                if getattr(shard, "strided_records", False):
                    shard.search_packed(q, k, filt, mask, after, out=buf, **ivf_kw)
                else:
                    rec = shard.search_packed(q, k, filt, mask, after, **ivf_kw) if (after is not None or ivf_kw) \
                        else shard.search_packed(q, k, filt, mask)
                    buf[:size].copy_(rec)
            except Exception as e:
                ok = failed("search", e)
                buf[size:].view(torch.int64).fill_(1)
            if self.world == 1:
                self._raise_if_failed([0 if ok else 1], "search")
                return shard.merge_packed(buf[:size], 1, nq, k)
            gathered = torch.empty((self.world, size + 8), dtype=torch.uint8, device=buf.device)
            dist.all_gather_into_tensor(gathered.view(-1), buf, group=self.group)
            self._raise_if_failed(gathered[:, size:].contiguous().view(torch.int64).cpu().numpy(), "search")
            if self.rank != 0:
                return None
            if getattr(shard, "strided_records", False):
                return shard.merge_packed(gathered.view(-1), self.world, nq, k, stride_bytes=size + 8)
            return shard.merge_packed(gathered[:, :size].contiguous().view(-1), self.world, nq, k)
        if op == OP_ADD:
            n, owner, gid_base, normalize = int(hdr[2]), int(hdr[3]), int(hdr[4]), bool(hdr[5])
            if owner != 0:   # the batch travels to its owner only
                if self.rank == 0:
                    self._send(vecs, owner)
                    self._send(tags, owner)
                elif self.rank == owner:
                    vecs = self._recv((n, self.dim), torch.float32, 0)
                    tags = self._recv((n,), torch.int32, 0)
            if self.rank == owner:
                ok = self._append(code, shard, vecs, tags, normalize, gid_base, n)
            self._verdict(ok, "add")
            return None
        if op == OP_ENCODE:
            # Data-parallel ingest ("chunks are independent => pure data-parallel, no collective; each
            # GPU appends its embeddings to its own shard"): rank 0 tokenised the texts and dealt them out in
            # batches of <= ENC_BATCH_SEQS sequences; every rank receives the round's token ids (one broadcast per
            # array: a 512-token chunk is 2 KB of ids against 4 KB of embedding), encodes ITS batches with its own
            # encoder and appends the rows to its own shard under the global ids rank 0 assigned.  No embedding
            # ever crosses ranks.  A sequence of length 0 is a blank text: a zero row (app/main.py:227-228).
            n_batches, first_owner, gid_base, normalize = int(hdr[2]), int(hdr[3]), int(hdr[4]), bool(hdr[5])
            total_seqs, total_tokens = int(hdr[6]), int(hdr[7])
            lens = self._bcast(enc[0] if enc else None, (total_seqs,), torch.int32)
            ids = self._bcast(enc[1] if enc else None, (max(total_tokens, 1),), torch.int32)
            tg = self._bcast(enc[2] if enc else None, (total_seqs,), torch.int32)
            try:
                lens_h = lens.cpu().numpy().astype(np.int64)
                tok0 = np.concatenate([[0], np.cumsum(lens_h)])
                for b in range(n_batches):
                    s0, s1 = b * ENC_BATCH_SEQS, min(total_seqs, (b + 1) * ENC_BATCH_SEQS)
                    if (first_owner + b) % self.world != self.rank:
                        continue
                    bl = lens_h[s0:s1]
                    out = np.zeros((s1 - s0, self.dim), dtype=np.float32)
                    keep = np.nonzero(bl > 0)[0]
                    if keep.size:
                        sub_ids = ids[int(tok0[s0]):int(tok0[s1])].cpu().numpy()
                        cu = np.concatenate([[0], np.cumsum(bl[keep])]).astype(np.int64)
                        try:
                            out[keep] = np.asarray(self.encoder().encode_flat(sub_ids, cu), dtype=np.float32)
                        except Exception as e:
                            # app/embedding_gen.py:165-170: an embedding error is printed and the text gets a zero vector
                            # (never a hit).  The rows are appended all the same: every rank's bookkeeping stays in step.
                            logger.error(f"[ERROR] encoder on rank {self.rank}: {e}; {int(keep.size)} texts get zero vectors")
                            out[:] = 0.0
                    if not self._append(code, shard, torch.from_numpy(out).to(self.device), tg[s0:s1].contiguous(), normalize,
                                        gid_base + s0, s1 - s0):
                        ok = False
            except Exception as e:
                ok = failed("encode", e)
            self._verdict(ok, "encode")
            return None
        if op == OP_IVF_BUILD:
            nlist, nprobe, dtype = int(hdr[2]), int(hdr[3]), {0: "f32", 1: "bf16", 2: "int8"}.get(int(hdr[4]), "f32")
            # the training is itself collective (all-reduce of the k-means sums over the data group): a rank that cannot
            # even start must not leave the others waiting in it, so every rank first says whether it can
            try:
                if not hasattr(shard, "build_ivf"):
                    raise RuntimeError("this shard type has no IVF")
            except Exception as e:
                ok = failed("ivf build", e)
            self._verdict(ok, "ivf build (precondition)")
            try:
                shard.build_ivf(nlist, nprobe, dtype, group=self.group if self.world > 1 else None)
            except Exception as e:
                ok = failed("ivf build", e)
            self._verdict(ok, "ivf build")
            return None
        if op == OP_DELETE:
            try:
                ordinal = self.extents[code].ordinal_of(int(hdr[2]))
                if ordinal is not None:
                    shard.delete(ordinal)
            except Exception as e:
                ok = failed("delete", e)
            self._verdict(ok, "delete")
            return None
        if op == OP_COUNT:
            try:
                vals = [int(shard.count), int(shard.rows), 0]
            except Exception as e:
                failed("count", e)
                vals = [0, 0, 1]
            t = torch.tensor(vals, dtype=torch.int64, device=self.device)     # the verdict rides on the reduction
            if self.world > 1:
                dist.all_reduce(t, group=self.group)
            live, rows, bad = (int(v) for v in t.cpu())
            if bad:
                raise CollectiveFailure(f"count failed on {bad} rank(s)")
            return [live, rows]
        if op == OP_GETROW:
            gid, owner = int(hdr[2]), int(hdr[3])
            row = None
            if self.rank == owner:
                try:
                    row = shard.get_row(self.extents[code].ordinal_of(gid))
                except Exception as e:
                    ok = failed("get_row", e)
                    row = torch.zeros((self.dim,), dtype=torch.float32, device=self.device)   # rank 0 is waiting for a row
                if owner != 0:
                    self._send(row, 0)
            elif self.rank == 0:
                row = self._recv((self.dim,), torch.float32, owner)
            self._verdict(ok, "get_row")
            return row.cpu().numpy() if (self.rank == 0 and row is not None) else None
        raise RuntimeError(f"unknown serving op {op}")

    def _append(self, code: int, shard, vecs: torch.Tensor, tags: torch.Tensor, normalize: bool, gid_base: int, n: int) -> bool:
        """The owner's append.  On failure nothing of the batch stays visible: rows a partial append may have published
        are tombstoned, the extent table is untouched."""
        before = None
        try:
            before = int(shard.rows)
            first = shard.add(vecs, tags, normalize, gid_base)
            self.extents[code].append(gid_base, first, n)
            return True
        except Exception as e:
            logger.error(f"append of {n} rows failed on rank {self.rank}: {e}")
            try:
                for o in range(before if before is not None else 0, int(shard.rows) if before is not None else 0):
                    shard.delete(o)
            except Exception:
                pass
            return False


def worker_loop(server: ShardServer) -> None:
    """Ranks 1..G-1: follow rank 0 until it shuts the service down."""
    while True:
        hdr = server.recv()
        if int(hdr[0]) == OP_SHUTDOWN:
            break
        try:
            server.execute(hdr)
        except CollectiveFailure as e:      # raised on every rank alike: rank 0 tells its caller, the service goes on
            logger.error(f"rank {server.rank}: {e}")
        # anything else (a communicator error, a bug) is not survivable in step with the others: it propagates, the process
        # exits non-zero and the launcher (torchrun) tears the job down and starts fresh processes
    dist.barrier(group=server.group)


class ShardedFront:
    """Rank 0: hands out ``ShardedIndex`` objects (the ``REGISTRY`` index factory) and owns the command stream.
    Every operation is serialised by one lock: the ranks must see one agreed order of collectives."""

    def __init__(self, server: ShardServer):
        assert server.rank == 0
        self.server = server
        self.lock = threading.RLock()
        self.indices: Dict[str, "ShardedIndex"] = {}
        self._closed = False

    def open_index(self, name: str) -> "ShardedIndex":
        with self.lock:
            idx = self.indices.get(name)
            if idx is None:
                code = len(self.indices)
                raw = np.frombuffer(name.encode("utf-8"), dtype=np.uint8)
                if raw.size > NAME_BYTES:
                    raise ValueError("index name too long")
                hdr = self.server.post([OP_OPEN, code, raw.size], raw)
                self.server.execute(hdr)         # CollectiveFailure: open on no rank, and not registered here
                idx = self.indices[name] = ShardedIndex(self, name, code)
            return idx

    def load_index(self, name: str, path: str) -> "ShardedIndex":
        """``IndexState.load``'s index loader: ``path`` is the manifest ``ShardedIndex.save`` wrote; every rank loads
        its own shard file.  Refused unless the world size equals the one that saved."""
        with open(path, encoding="utf-8") as f:
            man = json.load(f)
        s = self.server
        if man.get("format") != "rass-sharded-1" or int(man["world"]) != s.world:
            raise ValueError(f"{path}: saved by {man.get('world')} ranks, this service has {s.world}")
        with self.lock:
            if name in self.indices:
                raise ValueError(f"index {name!r} is already open")
            code = len(self.indices)
            base = os.path.join(os.path.dirname(path) or ".", man["base"])
            p = np.frombuffer(base.encode("utf-8"), dtype=np.uint8)
            nm = np.frombuffer(name.encode("utf-8"), dtype=np.uint8)
            runs = np.array([[g, r, n] for g, r, n in man["runs"]], dtype=np.int64).reshape(-1, 3)
            hdr = s.post([OP_LOAD, code, p.size, nm.size, runs.shape[0]], np.concatenate([p, nm]))
            s.execute(hdr, vecs=torch.from_numpy(runs if runs.size else np.zeros((1, 3), np.int64)))
            idx = self.indices[name] = ShardedIndex(self, name, code)
            idx._rows, idx._batches = int(man["rows"]), int(man["batches"])
            idx._deleted = set(int(x) for x in man["deleted"])
            for g, r, n in man["runs"]:
                if not idx._owner_rank or idx._owner_rank[-1] != int(r):
                    idx._owner_gid.append(int(g))
                    idx._owner_rank.append(int(r))
            idx._runs = [[int(g), int(r), int(n)] for g, r, n in man["runs"]]
            ivf = man.get("ivf") or {}
            idx._ivf_covered, idx._nprobe = int(ivf.get("covered", 0)), int(ivf.get("nprobe", 0))
            return idx

    def shutdown(self) -> None:
        """Clean collective exit: workers leave their loop, every rank meets in a barrier."""
        with self.lock:
            if self._closed:
                return
            self._closed = True
            self.server.post([OP_SHUTDOWN])
            dist.barrier(group=self.server.group)


class ShardedIndex:
    """``FlatIndex`` surface over all ranks' shards (what ``IndexState.index`` / ``HipIndexer`` talk to).
    Row ids are GLOBAL insertion ordinals, as on a single GPU."""

    def __init__(self, front: ShardedFront, name: str, code: int):
        self.front = front
        self.name = name
        self.code = code
        self.dim = front.server.dim
        self._rows = 0                      # global rows ever appended = next global id
        self._batches = 0                   # round-robin cursor
        self._owner_gid: List[int] = []     # extent table of the WHOLE index: sorted gid bases ...
        self._owner_rank: List[int] = []    # ... and the rank that holds each run
        self._runs: List[List[int]] = []    # every appended run (gid base, rank, rows): what a saved manifest keeps
        self._deleted = set()
        self._ivf_covered = 0               # global rows the shards' IVFs were built over (0 = no IVF)
        self._ivf_builds = 0
        self._nprobe = 0                    # lists per query and shard of an IVF probe (0 = the shards' default)

    # ---- bookkeeping
    @property
    def rows(self) -> int:
        return self._rows

    @property
    def count(self) -> int:
        """Live rows, reduced over the ranks (OpenSearchIndexer.has_any_data's count, app/main.py:1475)."""
        with self.front.lock:
            s = self.front.server
            live, _rows = s.execute(s.post([OP_COUNT, self.code]))
            return live

    @property
    def epoch(self) -> Tuple[int, int]:
        """(ids ever given out, rows tombstoned) from rank 0's own bookkeeping — no collective (``prefetch.index_epoch``)."""
        return self._rows, len(self._deleted), self._ivf_builds

    # ---- IVF (every shard builds its own over shared centroids; ivf.IvfPolicy decides when)
    def build_ivf(self, nlist: Optional[int] = None, nprobe: Optional[int] = None, dtype: Optional[str] = None) -> None:
        """Collective build of an IVF-``nlist`` on every shard (``OP_IVF_BUILD``); searches of k <= 32 then probe
        ``nprobe`` lists per shard + the rows appended since, and equal the flat result at nprobe = nlist."""
        from . import config
        nlist = int(nlist or config.RASS_IVF_NLIST)
        if nlist <= 0:
            raise ValueError("build_ivf needs nlist > 0 (RASS_IVF_NLIST)")
        from .ivf import SLAB_DTYPES
        code = SLAB_DTYPES[dtype or config.RASS_IVF_DTYPE]
        with self.front.lock:
            s = self.front.server
            self._nprobe = max(1, int(nprobe or config.RASS_IVF_NPROBE))
            s.execute(s.post([OP_IVF_BUILD, self.code, nlist, self._nprobe, code]))
            self._ivf_covered = self._rows
            self._ivf_builds += 1

    def _maybe_rebuild(self) -> None:
        """After an append: ``IvfPolicy`` on the GLOBAL row count (rank 0 decides for all shards)."""
        from . import config
        if config.RASS_IVF_NLIST <= 0:
            return
        from .ivf import IvfPolicy
        p = IvfPolicy.from_config()
        p.min_rows = max(p.min_rows, p.nlist * self.front.server.world)      # every shard needs nlist training rows
        if p.due(self._rows, self._ivf_covered):
            try:
                self.build_ivf()
            except Exception as e:      # the rows are in; searches stay exact (or on the old IVF + delta)
                logger.error(f"IVF (re)build of {self.name!r} failed: {e}")

    def _owner(self, gid: int) -> int:
        """The rank that holds global row ``gid``; -1 for a HOLE: an id an append consumed before it failed."""
        e = bisect.bisect_right(self._owner_gid, gid) - 1
        if e < 0 or not 0 <= gid < self._rows:
            raise IndexError(f"row {gid} out of range")
        return self._owner_rank[e]

    def _commit_run(self, owner: int, m: int) -> None:
        """Rank 0's bookkeeping for ``m`` global ids starting at the cursor: a run held by ``owner``, or a hole
        (``owner`` -1) when the append that was to store them failed — the cursor advances either way: once an ADD /
        ENCODE was posted a rank may have used those ids, and an id is never given out twice."""
        if not self._owner_rank or self._owner_rank[-1] != owner:
            self._owner_gid.append(self._rows)
            self._owner_rank.append(owner)
        self._runs.append([self._rows, owner, m])
        self._rows += m

    def _rollback(self, committed: List[Tuple[int, int]]) -> None:
        """A multi-part append failed half-way: tombstone the parts that went in (the caller registers none of the
        batch's documents, so those rows must not take slots in anyone's top-k).  Best effort."""
        for gid, m in committed:
            for g in range(gid, gid + m):
                try:
                    self.delete(g)
                except Exception as e:      # the service is degraded anyway; the caller hears about the original failure
                    logger.error(f"rollback of row {g} failed: {e}")
                    return

    # ---- write path
    def add(self, vecs: np.ndarray, tags: Optional[np.ndarray] = None, normalize: bool = True) -> int:
        """Append a batch; it goes to ONE rank (round-robin by batch, in chunks of <= 8192 rows).  Returns the
        global id of its first row."""
        v = np.ascontiguousarray(vecs, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}] vectors, got {v.shape}")
        n = v.shape[0]
        t = np.zeros(n, dtype=np.int32) if tags is None else np.ascontiguousarray(tags, dtype=np.int32)
        if t.shape != (n,) or (n and t.min() < 0):
            raise ValueError("tags must be one non-negative int32 per row")
        with self.front.lock:
            s = self.front.server
            first = self._rows
            if n == 0:
                return first
            owner = self._batches % s.world
            self._batches += 1
            committed: List[Tuple[int, int]] = []
            for a in range(0, n, ADD_CHUNK_ROWS):
                m = min(ADD_CHUNK_ROWS, n - a)
                dv = torch.from_numpy(v[a:a + m]).to(s.device)
                dt = torch.from_numpy(t[a:a + m]).to(s.device)
                hdr = s.post([OP_ADD, self.code, m, owner, self._rows, 1 if normalize else 0])
                try:
                    s.execute(hdr, dv, dt)
                except CollectiveFailure:
                    self._commit_run(-1, m)          # the owner stored nothing; the ids are burnt
                    self._rollback(committed)
                    raise
                committed.append((self._rows, m))
                self._commit_run(owner, m)           # only after the verdict
            self._maybe_rebuild()
            return first

    @property
    def can_encode(self) -> bool:
        """True when the ranks have encoders (``serving.start(..., encoder_factory=...)``): texts can be ingested
        data-parallel with ``add_texts`` instead of being embedded on rank 0 and shipped as vectors."""
        return self.front.server.encoder_factory is not None

    def add_texts(self, texts: List[str], tags: Optional[np.ndarray] = None, normalize: bool = True) -> int:
        """Embed ``texts`` on ALL ranks and append the rows (one row per text, in order; blank texts become zero
        rows as in app/main.py:227-228).  Rank 0 only tokenises; batches of <= ENC_BATCH_SEQS texts are dealt to
        the ranks round-robin and each rank encodes its batches into its own shard.  Returns the global id of the
        first row."""
        n = len(texts)
        t = np.zeros(n, dtype=np.int32) if tags is None else np.ascontiguousarray(tags, dtype=np.int32)
        if t.shape != (n,) or (n and t.min() < 0):
            raise ValueError("tags must be one non-negative int32 per row")
        with self.front.lock:
            s = self.front.server
            first = self._rows
            if n == 0:
                return first
            keep = [i for i, x in enumerate(texts) if x.strip()]
            lens = np.zeros(n, dtype=np.int32)
            flat = np.zeros(0, dtype=np.int32)
            if keep:
                ids, cu = s.encoder().tokenize([texts[i] for i in keep])
                flat = np.ascontiguousarray(ids, dtype=np.int32)
                lens[keep] = np.diff(np.asarray(cu, dtype=np.int64)).astype(np.int32)
            tok0 = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
            per_round = ENC_BATCH_SEQS * s.world          # one batch per rank and round
            committed: List[Tuple[int, int]] = []
            for a in range(0, n, per_round):
                b = min(n, a + per_round)
                n_batches = (b - a + ENC_BATCH_SEQS - 1) // ENC_BATCH_SEQS
                owner0 = self._batches % s.world
                r_ids = flat[int(tok0[a]):int(tok0[b])]
                hdr = s.post([OP_ENCODE, self.code, n_batches, owner0, self._rows, 1 if normalize else 0, b - a,
                              int(r_ids.size)])
                failed_ranks = None
                try:
                    s.execute(hdr, enc=(torch.from_numpy(lens[a:b].copy()),
                                        torch.from_numpy(r_ids.copy() if r_ids.size else np.zeros(1, np.int32)),
                                        torch.from_numpy(t[a:b].copy())))
                except CollectiveFailure as e:
                    failed_ranks, failure = set(e.failed_ranks), e
                # one batch per rank and round: a rank's status is its batch's.  Batches of the ranks that succeeded ARE
                # in their shards (under the ids assigned here) and are committed — and rolled back below when the round
                # failed; batches of failed ranks are holes.  The cursors advance in every case.
                for j in range(n_batches):
                    owner = (owner0 + j) % s.world
                    m = min(ENC_BATCH_SEQS, b - a - j * ENC_BATCH_SEQS)
                    if failed_ranks is not None and owner in failed_ranks:
                        self._commit_run(-1, m)
                    else:
                        committed.append((self._rows, m))
                        self._commit_run(owner, m)
                self._batches += n_batches
                if failed_ranks is not None:
                    self._rollback(committed)
                    raise failure
            self._maybe_rebuild()
            return first

    # ---- persistence (docstore.IndexState.save / load call these through the FlatIndex surface)
    def save(self, path: str) -> None:
        """Every rank saves its shard to ``<base>.shard<r>of<G>`` (base = ``path`` without a trailing ``.tmp``);
        when ALL are durable rank 0 writes the manifest — world size, the run table (gid base, rank, rows) the
        ranks' extent tables are rebuilt from, tombstones — to ``path``, which the caller renames into place."""
        base = path[:-4] if path.endswith(".tmp") else path
        with self.front.lock:
            s = self.front.server
            raw = np.frombuffer(base.encode("utf-8"), dtype=np.uint8)
            if raw.size > s.payload_bytes:
                raise ValueError("path too long")
            s.execute(s.post([OP_SAVE, self.code, raw.size], raw))
            man = {"format": "rass-sharded-1", "world": s.world, "base": os.path.basename(base), "rows": self._rows,
                   "batches": self._batches, "runs": self._runs, "deleted": sorted(self._deleted),
                   "ivf": {"covered": self._ivf_covered, "nprobe": self._nprobe}}
            with open(path, "w", encoding="utf-8") as f:
                json.dump(man, f)
                f.flush()
                os.fsync(f.fileno())

    def saved_files(self, manifest_path: str) -> List[str]:
        """The shard files a manifest of this index names (``IndexState.save`` removes the previous generation's)."""
        try:
            with open(manifest_path, encoding="utf-8") as f:
                man = json.load(f)
            d = os.path.dirname(manifest_path) or "."
            files = [ShardServer.shard_file(os.path.join(d, man["base"]), r, int(man["world"])) for r in range(int(man["world"]))]
            return files + [f + ".ivf" for f in files] + [f + ".ext" for f in files]   # IVF + extent table next to the rows
        except (OSError, ValueError, KeyError):
            return []

    def delete(self, row: int) -> None:
        with self.front.lock:
            self._owner(row)                 # range check
            if row in self._deleted:
                return
            s = self.front.server
            s.execute(s.post([OP_DELETE, self.code, int(row)]))
            self._deleted.add(row)

    def get_row(self, row: int) -> np.ndarray:
        with self.front.lock:
            s = self.front.server
            owner = self._owner(row)
            if owner < 0:
                raise IndexError(f"row {row} was never stored (the append that consumed its id failed)")
            return s.execute(s.post([OP_GETROW, self.code, int(row), owner]))

    def get_rows(self, first_row: int, n: int) -> np.ndarray:
        """Stored (normalised) rows [first_row, first_row + n) by global id, one OP_GETROW each: what
        ``docstore.IndexState.save_delta`` reads back for its segment (O(delta) rows); a hole (a failed append) comes back as a
        zero row — its doc slot is None and the replay tombstones it."""
        out = np.zeros((int(n), self.dim), dtype=np.float32)
        for i in range(int(n)):
            try:
                out[i] = np.asarray(self.get_row(int(first_row) + i), dtype=np.float32).reshape(-1)
            except IndexError:
                pass
        return out

    # ---- read path
    def search(self, queries: np.ndarray, k: int, q_filter: Optional[np.ndarray] = None,
               q_filter_mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] queries, got {q.shape}")
        k = int(k)
        if k < 1:
            raise ValueError("k must be >= 1")
        if q_filter_mask is not None and q_filter is None:
            raise ValueError("q_filter_mask needs q_filter")
        nq = q.shape[0]
        out_s = np.full((nq, k), -np.inf, dtype=np.float32)
        out_i = np.full((nq, k), -1, dtype=np.int64)
        s = self.front.server
        off = MAX_Q * self.dim * 4
        for a in range(0, nq, MAX_Q):
            b = min(MAX_Q, nq - a)
            payload = np.zeros(s.payload_bytes, dtype=np.uint8)
            payload[:b * self.dim * 4] = q[a:a + b].view(np.uint8).reshape(-1)
            flags = 0
            if q_filter is not None:
                flags |= 1
                payload[off:off + b * 4] = np.ascontiguousarray(q_filter[a:a + b], dtype=np.int32).view(np.uint8)
            if q_filter_mask is not None:
                flags |= 2
                payload[off + MAX_Q * 4:off + MAX_Q * 4 + b * 4] = \
                    np.ascontiguousarray(q_filter_mask[a:a + b], dtype=np.int32).view(np.uint8)
            # k > 32: passes of <= 32; pass p asks every shard for its best rows strictly behind pass p-1's last
            # global hit (score desc, id asc) — the single-index multipass of rass_index_search_ex, across shards
            done = 0
            if k > MAX_K:
                flags |= SEARCH_EXACT                    # every pass of a deep search scans the flat shards
            while done < k:
                kk = min(MAX_K, k - done)
                pf = flags
                if done > 0:
                    pf |= 4
                    last_s = out_s[a:a + b, done - 1].copy()
                    last_i = out_i[a:a + b, done - 1].copy()
                    exhausted = last_i < 0
                    last_s[exhausted] = -np.inf          # nothing ranks behind -inf: an exhausted query stays empty
                    payload[off + 2 * MAX_Q * 4:off + 2 * MAX_Q * 4 + b * 4] = last_s.view(np.uint8)
                    payload[off + 3 * MAX_Q * 4:off + 3 * MAX_Q * 4 + b * 8] = last_i.view(np.uint8)
                with self.front.lock:
                    # an unfiltered first pass moves only its queries; filters / continuation bounds sit behind them
                    sc, ids = s.execute(s.post([OP_SEARCH, self.code, b, kk, pf, self._nprobe], payload,
                                               payload_len=b * self.dim * 4 if pf == 0 else None))
                out_s[a:a + b, done:done + kk] = sc
                out_i[a:a + b, done:done + kk] = ids
                done += kk
                if np.all(ids[:, -1] < 0):               # every query of the group ran out of rows
                    break
        return out_s, out_i


def start(shard_factory: Callable[[str], object], dim: int, device: Optional[torch.device] = None,
          group: Optional[dist.ProcessGroup] = None, install_registry: bool = True,
          encoder_factory: Optional[Callable[[], object]] = None,
          shard_loader: Optional[Callable[[str, str], object]] = None) -> Optional[ShardedFront]:
    """Call on EVERY rank after ``init_process_group``.  Rank 0 gets the front back at once (and, with
    ``install_registry``, ``docstore.REGISTRY`` now opens sharded indices, so ``HipIndexer`` /
    ``store_fhir_docs_in_opensearch`` serve the multi-GPU index unchanged); the other ranks stay inside
    this call, following rank 0, until it calls ``front.shutdown()``, and then return ``None``."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    server = ShardServer(shard_factory, dim, device, group, encoder_factory, shard_loader)
    if server.rank != 0:
        worker_loop(server)
        return None
    front = ShardedFront(server)
    if install_registry:
        from .docstore import REGISTRY
        REGISTRY.set_index_factory(front.open_index)
    return front


def hip_encoder_factory(model_dir: str, device_index: int) -> Callable[[], object]:
    """One sentence encoder per process per GPU (weights loaded on first use), for ``start(encoder_factory=...)``."""
    def make():
        from .encoder import HipSentenceEncoder
        return HipSentenceEncoder.from_dir(model_dir, device=device_index)
    return make


def hip_shard_loader(device_index: int, dim: int) -> Callable[[str, str], "HipServingShard"]:
    """``start(shard_loader=...)`` for HIP shards: ``rass_index_load`` of the rank's shard file."""
    from .engine import Engine

    def load(name: str, path: str) -> HipServingShard:
        from .ivf import IvfBackedIndex, IvfPolicy
        from . import config
        idx = IvfBackedIndex.load(Engine.get(device_index, dim), name, path, IvfPolicy.manual(config.RASS_IVF_NPROBE))
        if config.RASS_PREFILTER != "off":
            idx.set_prefilter(config.RASS_PREFILTER)
        return HipServingShard(idx)
    return load


def hip_shard_factory(device_index: int, dim: int) -> Callable[[str], HipServingShard]:
    """The production shard factory: one engine per process per GPU."""
    from .engine import Engine

    def make(name: str) -> HipServingShard:
        # an IvfBackedIndex with no policy of its own: it is a plain flat shard until rank 0 posts OP_IVF_BUILD
        from .ivf import IvfBackedIndex, IvfPolicy
        from . import config
        idx = IvfBackedIndex(Engine.get(device_index, dim).open_index(name), IvfPolicy.manual(config.RASS_IVF_NPROBE))
        if config.RASS_PREFILTER != "off":      # every shard's searches of k <= 16: int8 / bf16 candidates + exact re-rank
            idx.set_prefilter(config.RASS_PREFILTER)
        return HipServingShard(idx)
    return make
