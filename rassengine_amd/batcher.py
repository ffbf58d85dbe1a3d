"""Cross-request query micro-batching (SURVEY §8b "Threading", §8f-2).

The reference answers one ``/ask`` at a time with one k-NN request each
(app/main.py:2878-2885 -> 1552).  The fused scan costs the same HBM pass for 1 or 32
queries, so concurrent requests are coalesced: callers ``await batcher.search(...)``; a
single drain task collects up to ``max_batch`` (<= 32) pending queries — waiting at most
``max_delay_ms`` after the first one — and issues ONE scan for all of them.  Requests with
different ``k`` share a scan at the largest ``k`` and are sliced afterwards (a prefix of a
top-k list under a total order is the top-k' list).
"""
from __future__ import annotations

import asyncio
from typing import List, Optional, Tuple

import numpy as np


class QueryBatcher:
    def __init__(self, index, max_batch: int = 32, max_delay_ms: float = 0.25):
        if not 1 <= max_batch <= 32:
            raise ValueError("max_batch must be in [1, 32] (one scan launch)")
        self.index = index                  # FlatIndex-like: search(queries, k, q_filter)
        self.max_batch = max_batch
        self.max_delay = max_delay_ms / 1e3
        self._queue: Optional[asyncio.Queue] = None
        self._task: Optional[asyncio.Task] = None
        self.scans = 0                      # statistics: scans issued / queries served
        self.served = 0

    def _ensure_started(self) -> None:
        loop = asyncio.get_running_loop()
        if self._task is None or self._task.done() or self._task.get_loop() is not loop:
            self._queue = asyncio.Queue()
            self._task = loop.create_task(self._drain())

    async def search(self, query: np.ndarray, k: int, filter_value: int = -1, filter_mask: int = -1
                     ) -> Tuple[np.ndarray, np.ndarray]:
        """One query vector [dim] -> (scores [k], ids [k]).  ``filter_value`` < 0 = no filter; else rows
        with ``(tag & filter_mask) == filter_value`` (mask -1 = exact compare)."""
        self._ensure_started()
        fut = asyncio.get_running_loop().create_future()
        await self._queue.put((np.asarray(query, dtype=np.float32).reshape(-1), int(k),
                               (int(filter_value), int(filter_mask)), fut))
        return await fut

    async def _drain(self) -> None:
        q = self._queue
        while True:
            first = await q.get()
            batch = [first]
            loop = asyncio.get_running_loop()
            deadline = loop.time() + self.max_delay
            while len(batch) < self.max_batch:
                timeout = deadline - loop.time()
                if timeout <= 0:
                    break
                try:
                    batch.append(await asyncio.wait_for(q.get(), timeout))
                except asyncio.TimeoutError:
                    break
            await self._run(batch)

    async def _run(self, batch: List[tuple]) -> None:
        try:
            qs = np.stack([b[0] for b in batch])
            kmax = max(b[1] for b in batch)
            codes = np.array([b[2][0] for b in batch], dtype=np.int64)
            masks = np.array([b[2][1] if b[2][0] >= 0 else 0 for b in batch], dtype=np.int64)
            if not bool((codes >= 0).any()):
                scores, ids = await asyncio.to_thread(self.index.search, qs, kmax)
            elif bool((masks[codes >= 0] == -1).all()):          # plain exact filters: the plain kernel variant
                scores, ids = await asyncio.to_thread(self.index.search, qs, kmax, codes.astype(np.int32))
            else:
                scores, ids = await asyncio.to_thread(self.index.search, qs, kmax, codes.astype(np.int32),
                                                      (masks & 0xFFFFFFFF).astype(np.uint32).view(np.int32))
            self.scans += 1
            self.served += len(batch)
            for i, (_, k, _, fut) in enumerate(batch):
                if not fut.done():
                    fut.set_result((scores[i, :k].copy(), ids[i, :k].copy()))
        except Exception as e:  # every waiter sees the failure
            for (_, _, _, fut) in batch:
                if not fut.done():
                    fut.set_exception(e)

    async def close(self) -> None:
        if self._task is not None:
            self._task.cancel()
            try:
                await self._task
            except (asyncio.CancelledError, Exception):
                pass
            self._task = None


class CrossIndexBatcher:
    """The same coalescing ACROSS indices of one engine: the reference keeps one index per user
    (app/main.py:346-347), so concurrent users never hit the same index and a per-index batcher never has two
    requests to share.  Requests (index, query, k, filter) of any indices are collected for ``max_delay_ms`` and
    answered by ONE cross-index scan (``Engine.search_multi`` -> ``rass_index_search_multi``): every distinct index
    of the batch is streamed once, in the same launch as the others."""

    def __init__(self, engine, max_batch: int = 32, max_delay_ms: float = 0.25):
        if not 1 <= max_batch <= 32:
            raise ValueError("max_batch must be in [1, 32] (one scan launch)")
        self.engine = engine
        self.max_batch = max_batch
        self.max_delay = max_delay_ms / 1e3
        self._queue: Optional[asyncio.Queue] = None
        self._task: Optional[asyncio.Task] = None
        self.scans = 0
        self.served = 0

    def _ensure_started(self) -> None:
        loop = asyncio.get_running_loop()
        if self._task is None or self._task.done() or self._task.get_loop() is not loop:
            self._queue = asyncio.Queue()
            self._task = loop.create_task(self._drain())

    async def search(self, index, query: np.ndarray, k: int, filter_value: int = -1, filter_mask: int = -1
                     ) -> Tuple[np.ndarray, np.ndarray]:
        self._ensure_started()
        fut = asyncio.get_running_loop().create_future()
        await self._queue.put((index, np.asarray(query, dtype=np.float32).reshape(-1), int(k),
                               (int(filter_value), int(filter_mask)), fut))
        return await fut

    async def _drain(self) -> None:
        q = self._queue
        while True:
            batch = [await q.get()]
            loop = asyncio.get_running_loop()
            deadline = loop.time() + self.max_delay
            while len(batch) < self.max_batch:
                timeout = deadline - loop.time()
                if timeout <= 0:
                    break
                try:
                    batch.append(await asyncio.wait_for(q.get(), timeout))
                except asyncio.TimeoutError:
                    break
            await self._run(batch)

    async def _run(self, batch: List[tuple]) -> None:
        try:
            qs = np.stack([b[1] for b in batch])
            kmax = min(32, max(b[2] for b in batch))
            codes = np.array([b[3][0] for b in batch], dtype=np.int64)
            masks = np.array([b[3][1] if b[3][0] >= 0 else 0 for b in batch], dtype=np.int64)
            f = m = None
            if bool((codes >= 0).any()):
                f = codes.astype(np.int32)
                if not bool((masks[codes >= 0] == -1).all()):
                    m = (masks & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
            scores, ids = await asyncio.to_thread(self.engine.search_multi, [b[0] for b in batch], qs, kmax, f, m)
            self.scans += 1
            self.served += len(batch)
            for i, (_, _, k, _, fut) in enumerate(batch):
                if not fut.done():
                    fut.set_result((scores[i, :k].copy(), ids[i, :k].copy()))
        except Exception as e:
            for b in batch:
                if not b[4].done():
                    b[4].set_exception(e)

    async def close(self) -> None:
        if self._task is not None:
            self._task.cancel()
            try:
                await self._task
            except (asyncio.CancelledError, Exception):
                pass
            self._task = None
