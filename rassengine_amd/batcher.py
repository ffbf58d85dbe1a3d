"""Cross-request query micro-batching (SURVEY §8b "Threading", §8f-2).

The reference answers one ``/ask`` at a time with one k-NN request each
(app/main.py:2878-2885 -> 1552).  The fused scan costs the same HBM pass for 1 or 32
queries, so concurrent requests are coalesced: callers ``await batcher.search(...)``; a
single drain task collects up to ``max_batch`` (<= 32) pending queries — everything queued
when it runs; optionally lingering ``max_delay_ms`` — and issues ONE scan for all of them.  Requests with
different ``k`` share a scan at the largest ``k`` and are sliced afterwards (a prefix of a
top-k list under a total order is the top-k' list).
"""
from __future__ import annotations

import asyncio
from typing import List, Optional, Tuple

import numpy as np


async def _collect(q: "asyncio.Queue", max_batch: int, max_delay: float) -> List[tuple]:
    """The next batch of a drain task: the first entry, then whatever else is ALREADY queued or becomes queued by
    tasks that are runnable right now (one ``sleep(0)``: requests that left one encoder forward together wake in the
    same loop iteration and all enqueue before the scan is issued).  No timer by default — while a scan runs, arrivals
    pile up and form the next batch; asyncio timers on the selector loop round up to whole milliseconds, which is
    longer than a 1 M-row scan.  ``max_delay`` > 0 additionally lingers that long for a batch that is not full."""
    batch = [await q.get()]
    await asyncio.sleep(0)
    while len(batch) < max_batch and not q.empty():
        batch.append(q.get_nowait())
    if max_delay > 0 and len(batch) < max_batch:
        loop = asyncio.get_running_loop()
        deadline = loop.time() + max_delay
        while len(batch) < max_batch:
            timeout = deadline - loop.time()
            if timeout <= 0:
                break
            try:
                batch.append(await asyncio.wait_for(q.get(), timeout))
            except asyncio.TimeoutError:
                break
            while len(batch) < max_batch and not q.empty():
                batch.append(q.get_nowait())
    return batch


class QueryBatcher:
    def __init__(self, index, max_batch: int = 32, max_delay_ms: float = 0.0):
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
            batch = await _collect(q, self.max_batch, self.max_delay)
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

    def __init__(self, engine, max_batch: int = 32, max_delay_ms: float = 0.0):
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
            batch = await _collect(q, self.max_batch, self.max_delay)
            await self._run(batch)

    MAX_TILES = 65536     # rass_index_search_multi's budget: 32-row tiles over the distinct indices of one call

    @staticmethod
    def _filters(entries):
        codes = np.array([e[3][0] for e in entries], dtype=np.int64)
        masks = np.array([e[3][1] if e[3][0] >= 0 else 0 for e in entries], dtype=np.int64)
        f = m = None
        if bool((codes >= 0).any()):
            f = codes.astype(np.int32)
            if not bool((masks[codes >= 0] == -1).all()):
                m = (masks & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        return f, m

    def _plan(self, batch: List[tuple]):
        """Split a drained batch into calls the engine accepts.  ``multi`` groups: entries whose DISTINCT indices fit
        the cross-index tile budget together (``index.multi_tiles``: 0 = this index cannot join — bf16 corpus,
        caller-assigned ids, more than half the budget on its own); ``solo``: per index, entries answered by that
        index's own ``search`` (one scan for all of them).  Nobody's request fails because of who else is in the batch."""
        multi, solo = [], {}
        cur, cur_tiles, cur_seen = [], 0, {}
        for e in batch:
            ix = e[0]
            tiles = int(getattr(ix, "multi_tiles", 0) or 0)
            if tiles <= 0:
                solo.setdefault(id(ix), []).append(e)
                continue
            extra = 0 if id(ix) in cur_seen else tiles
            if cur and cur_tiles + extra > self.MAX_TILES:
                multi.append(cur)
                cur, cur_tiles, cur_seen = [], 0, {}
                extra = tiles
            cur.append(e)
            cur_tiles += extra
            cur_seen[id(ix)] = True
        if cur:
            multi.append(cur)
        # a group whose entries all name ONE index gains nothing from the work-list launch: that index's own search
        # is the tuned path (sample floor, no per-call tile list) and answers bit-identically
        kept = []
        for group in multi:
            if len({id(e[0]) for e in group}) == 1:
                solo.setdefault(id(group[0][0]), []).extend(group)
            else:
                kept.append(group)
        return kept, list(solo.values())

    @staticmethod
    def _answer(entries, scores, ids) -> None:
        for i, (_, _, k, _, fut) in enumerate(entries):
            if not fut.done():
                fut.set_result((scores[i, :k].copy(), ids[i, :k].copy()))

    async def _run_solo(self, entries) -> None:
        """One index's entries through its own search (any dtype / id scheme / size)."""
        try:
            qs = np.stack([e[1] for e in entries])
            kmax = max(e[2] for e in entries)
            f, m = self._filters(entries)
            if f is None:
                scores, ids = await asyncio.to_thread(entries[0][0].search, qs, kmax)
            elif m is None:
                scores, ids = await asyncio.to_thread(entries[0][0].search, qs, kmax, f)
            else:
                scores, ids = await asyncio.to_thread(entries[0][0].search, qs, kmax, f, m)
            self.scans += 1
            self._answer(entries, scores, ids)
        except Exception as e:
            for en in entries:
                if not en[4].done():
                    en[4].set_exception(e)

    async def _run(self, batch: List[tuple]) -> None:
        multi, solo = self._plan(batch)
        for entries in multi:
            try:
                qs = np.stack([e[1] for e in entries])
                kmax = min(32, max(e[2] for e in entries))
                f, m = self._filters(entries)
                scores, ids = await asyncio.to_thread(self.engine.search_multi, [e[0] for e in entries], qs, kmax, f, m)
                self.scans += 1
                self._answer(entries, scores, ids)
            except Exception:
                # the engine refused the group (e.g. an index grew past the budget since it was planned): every index
                # of the group is served on its own instead
                per_index = {}
                for e in entries:
                    per_index.setdefault(id(e[0]), []).append(e)
                for group in per_index.values():
                    await self._run_solo(group)
        for entries in solo:
            await self._run_solo(entries)
        self.served += len(batch)

    async def close(self) -> None:
        if self._task is not None:
            self._task.cancel()
            try:
                await self._task
            except (asyncio.CancelledError, Exception):
                pass
            self._task = None


class EmbedBatcher:
    """Coalesces concurrent EMBED requests into one varlen encoder forward.

    The reference keeps up to ``MAX_EMBED_CONCURRENCY`` = 5 embed requests in flight, one text each
    (app/main.py:250-260), and every ``/ask`` awaits ``embed_query`` (app/main.py:2800) — the first place where an
    unmodified ``ask()`` yields the event loop on this path, i.e. where the EMBEDS of different users can meet (their
    k-NN scans meet at the next await, ``ensure_index_exists``: prefetch.py).
    A forward of 32 short queries costs about as much as a forward of one (the few-rows GEMMs stream the same
    weights), so callers enqueue their texts and ONE worker thread — which owns every encoder call of the
    process — takes whatever has arrived, up to ``max_seqs`` sequences, and runs one ``encode`` for all of it;
    the rows go back to their futures on the loops they came from.

    Who waits for what: while a forward runs, arrivals pile up and form the next batch (no timer involved).  When
    the previous forward had company (more than one entry) the worker also lingers until the queue has been quiet
    for ``quiet_us`` (default 50 us), never longer than ``max_delay_ms`` (default 0.2 ms) after the first arrival;
    a lone caller in a quiet process — the previous forward served one entry — is not held back at all.  (The linger
    is a ``threading.Condition`` wait: asyncio timers on the selector loop round up to whole milliseconds.)  Entries of
    more than ``max_seqs`` texts — upload slices — run on their own, behind any waiting small entries, so a query never
    queues behind more than one upload slice; a waiting slice is served after at most ``big_after`` (8) small batches
    in a row, so queries cannot starve an upload either.

    Errors are per entry, as the reference's are per text: if a coalesced forward fails, every entry is retried
    on its own and only the ones that fail again see the exception.
    """

    def __init__(self, encode, max_seqs: int = 64, max_delay_ms: float = 0.2, quiet_us: float = 50.0):
        import collections
        import threading
        if max_seqs < 1:
            raise ValueError("max_seqs must be >= 1")
        self._encode = encode               # callable: List[str] -> np.ndarray [n, dim] (runs on the worker thread)
        self.max_seqs = int(max_seqs)
        self.max_delay = max(0.0, float(max_delay_ms)) / 1e3
        self.quiet = max(0.0, float(quiet_us)) / 1e6
        self._cv = threading.Condition()
        self._small = collections.deque()   # entries (texts, loop, future) with <= max_seqs texts
        self._big = collections.deque()
        self._arrivals = 0                  # bumped on every submit (the linger watches it)
        self._stop = False
        self._thread: Optional["threading.Thread"] = None
        self.forwards = 0                   # encode() calls issued for coalesced batches
        self.served = 0                     # entries answered
        self.retries = 0                    # entries re-run alone after a failed batch
        self._company = False               # the previous forward served more than one entry
        self.big_after = 8                  # small batches served in a row while an upload slice waits
        self._small_streak = 0

    # ------------------------------------------------------------------ caller side (any event loop)
    async def embed(self, texts: List[str]) -> np.ndarray:
        """fp32 [len(texts), dim] for NON-BLANK texts, in order."""
        if not texts:
            raise ValueError("embed() needs at least one text")
        loop = asyncio.get_running_loop()
        fut = loop.create_future()
        entry = (list(texts), loop, fut)
        with self._cv:
            if self._stop:
                raise RuntimeError("EmbedBatcher is closed")
            if self._thread is None:
                import threading
                self._thread = threading.Thread(target=self._worker, name="rass-embed-batcher", daemon=True)
                self._thread.start()
            (self._small if len(entry[0]) <= self.max_seqs else self._big).append(entry)
            self._arrivals += 1
            self._cv.notify()
        return await fut

    # ------------------------------------------------------------------ worker side
    def _take(self):
        """Called with the lock held and at least one entry queued: the entries of the next forward.  Small entries
        (queries) go first, but a waiting upload slice is served after at most ``big_after`` consecutive small
        batches: sustained query traffic cannot starve an upload."""
        if self._big and (not self._small or self._small_streak >= self.big_after):
            self._small_streak = 0
            return [self._big.popleft()]
        self._small_streak = self._small_streak + 1 if self._big else 0
        batch, total = [], 0
        while self._small and total + len(self._small[0][0]) <= self.max_seqs:
            e = self._small.popleft()
            batch.append(e)
            total += len(e[0])
        return batch

    def _worker(self) -> None:
        import time
        while True:
            with self._cv:
                while not self._small and not self._big and not self._stop:
                    self._cv.wait()
                if self._stop and not self._small and not self._big:
                    return
                if self._small and self.max_delay > 0 and self._company:
                    # linger: until quiet for `quiet`, at most `max_delay` after the first arrival, or the cap is met
                    t_end = time.perf_counter() + self.max_delay
                    while sum(len(e[0]) for e in self._small) < self.max_seqs and not self._stop:
                        seen = self._arrivals
                        left = t_end - time.perf_counter()
                        if left <= 0:
                            break
                        self._cv.wait(min(self.quiet, left))
                        if self._arrivals == seen:
                            break
                batch = self._take()
                self._company = len(batch) > 1
            self._run(batch)

    @staticmethod
    def _deliver(entries_results) -> None:
        """Results / exceptions back to their futures, one thread-safe call per event loop."""
        by_loop = {}
        for (texts, loop, fut), res in entries_results:
            by_loop.setdefault(loop, []).append((fut, res))

        def settle(pairs):
            for fut, res in pairs:
                if fut.done():          # the caller was cancelled
                    continue
                if isinstance(res, BaseException):
                    fut.set_exception(res)
                else:
                    fut.set_result(res)

        for loop, pairs in by_loop.items():
            try:
                loop.call_soon_threadsafe(settle, pairs)
            except RuntimeError:        # that loop is closed: nobody is waiting any more
                pass

    def _run(self, batch) -> None:
        texts = [t for e in batch for t in e[0]]
        try:
            vecs = np.asarray(self._encode(texts), dtype=np.float32)
            if vecs.ndim != 2 or vecs.shape[0] != len(texts):
                raise RuntimeError(f"encoder returned shape {vecs.shape} for {len(texts)} texts")
            self.forwards += 1
            out, pos = [], 0
            for e in batch:
                n = len(e[0])
                out.append((e, np.array(vecs[pos:pos + n], dtype=np.float32, order="C")))
                pos += n
        except BaseException as ex:  # noqa: BLE001 - every waiter must hear about it
            if len(batch) == 1:
                out = [(batch[0], ex)]
            else:
                out = []
                for e in batch:
                    self.retries += 1
                    try:
                        out.append((e, np.array(self._encode(e[0]), dtype=np.float32, order="C")))
                    except BaseException as ex1:  # noqa: BLE001
                        out.append((e, ex1))
        self.served += len(batch)
        self._deliver(out)

    def close(self) -> None:
        """Stops the worker after it has answered what is queued."""
        with self._cv:
            self._stop = True
            self._cv.notify_all()
            t = self._thread
        if t is not None:
            t.join(timeout=30)
