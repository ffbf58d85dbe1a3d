"""The k-NN half of concurrent ``/ask`` requests, coalesced WITHOUT touching the caller.

``ask()`` (app/main.py:2800-2802; the websocket route 3119-3121) runs

    query_emb = await embed_query(query)                 # await #1 — EmbedBatcher coalesces the embeds here
    await ensure_index_exists(os_client, index_name)     # await #2 — owned by the shim, same asyncio task
    os_indexer = OpenSearchIndexer(os_client, index_name)
    ... os_indexer.semantic_search(query_emb=..., k=top_k, ...)   # SYNCHRONOUS (2878-2885 -> 1552)

The search itself is a synchronous call, so the requests of 32 users cannot meet there; but at await #2 the same
task has just produced the query embedding and now names the index — everything a scan needs.  So:

* ``embed_query`` REMEMBERS (current task -> a private copy of the embedding) (``remember``);
* ``ensure_index_exists`` (``run``) enqueues that embedding on the index's batcher for an UNFILTERED top-32
  (``K_PREFETCH``), AWAITS it — this is where the requests of different users share one scan launch — and parks
  (scores, row ids, the index state, its epoch) on the task;
* the synchronous ``semantic_search`` / ``hybrid_*`` / ``multi_intent_search`` (``take``) answer from the parked list
  when that is EXACT: same embedding bytes, same index state object, same epoch (rows appended and rows tombstoned,
  both monotonic — a write landing between the prefetch and its use falls back), k <= 32, and either no filter or at
  least k of the 32 hits pass the patient / doc_type filter: every matching row outside the unfiltered top-32 ranks
  below all 32 under (score desc, id asc), so the first k matching hits ARE the pre-filtered top-k.  Otherwise the
  caller runs today's inline scan.  Scores are the same kernel's per-row dot products either way (a prefix of a
  top-32 list under a total order is the top-k list), so the answer is bit-identical to the serial path.

A request that is ALONE in the process (no other embed in flight, nothing waiting on the batcher) skips the prefetch
and scans inline exactly as before: no added latency for a lone user, no wasted scan for the seven intents that never
read the vector.  ``RASS_KNN_PREFETCH``: 0 = off, 1 = when there is company (default), 2 = always.
"""
from __future__ import annotations

import asyncio
import logging
import threading
import weakref
from typing import Any, Optional, Tuple

import numpy as np

from . import config

logger = logging.getLogger("rassengine_amd")

K_PREFETCH = 32          # one scan launch serves k <= 32; deeper lists leave room for filters
K_PREFETCH_APPROX = 16   # an index in a prefilter mode (engine.set_prefilter: int8 / bf16 candidates + exact re-rank) serves k <= 16
                         # from its candidate scan — a quarter / half of the bytes per launch; deeper lists take the exact scan


class _Slot:
    __slots__ = ("emb", "pending", "st", "epoch", "scores", "ids", "k", "approx")

    def __init__(self, emb: np.ndarray):
        self.emb = emb
        self.pending = True
        self.st = None
        self.epoch = None
        self.scores = None
        self.ids = None
        self.k = K_PREFETCH
        self.approx = False


_slots: "weakref.WeakKeyDictionary[asyncio.Task, _Slot]" = weakref.WeakKeyDictionary()
_lock = threading.Lock()
_embeds_in_flight = 0    # embed_query calls between entry and return, over every event loop of the process
_waiting = 0             # prefetches enqueued on a batcher and not answered yet
stats = {"prefetched": 0, "alone": 0, "answered": 0, "stale": 0, "filter_short": 0, "other_query": 0, "failed": 0}


def reset_stats() -> None:
    for key in stats:
        stats[key] = 0


def _task() -> Optional["asyncio.Task"]:
    try:
        return asyncio.current_task()
    except RuntimeError:        # no running loop: a synchronous caller outside any coroutine
        return None


def embed_enter() -> None:
    global _embeds_in_flight
    with _lock:
        _embeds_in_flight += 1


def embed_exit() -> None:
    global _embeds_in_flight
    with _lock:
        _embeds_in_flight -= 1


def remember(emb: np.ndarray) -> None:
    """``embed_query`` hands the embedding it is about to return to the task that asked for it."""
    if config.RASS_KNN_PREFETCH <= 0:
        return
    t = _task()
    if t is not None and emb is not None and np.size(emb):
        _slots[t] = _Slot(np.array(emb, dtype=np.float32, copy=True).reshape(-1))


def index_epoch(index: Any) -> Tuple[int, int]:
    """(rows ever appended, rows tombstoned) — both only grow, so equal epochs mean no write in between."""
    ep = getattr(index, "epoch", None)
    if ep is not None:
        return ep
    rows = int(index.rows)
    return rows, rows - int(index.count)


def _batcher_for(st) -> Any:
    """The engine-wide cross-index batcher for an index of the HIP engine (concurrent users have one index each,
    app/main.py:346-347); the state's own ``QueryBatcher`` for anything else (sharded fronts, test doubles)."""
    eng = getattr(st.index, "engine", None)
    if eng is not None and hasattr(eng, "search_multi"):
        b = getattr(eng, "_cross_batcher", None)
        if b is None:
            from .batcher import CrossIndexBatcher
            b = eng._cross_batcher = CrossIndexBatcher(eng)
        return b, True
    if st.batcher is None:
        from .batcher import QueryBatcher
        st.batcher = QueryBatcher(st.index)
    return st.batcher, False


async def run(st) -> None:
    """``ensure_index_exists``'s part: never raises, never changes what the caller will see."""
    global _waiting
    mode = config.RASS_KNN_PREFETCH
    if mode <= 0 or st is None:
        return
    t = _task()
    slot = _slots.get(t) if t is not None else None
    if slot is None or not slot.pending:
        return
    slot.pending = False            # one prefetch per embed_query: a later ensure_index_exists of this task is an upload's
    try:
        if mode == 1 and _embeds_in_flight <= 0 and _waiting <= 0:
            stats["alone"] += 1
            return
        index = st.index
        if slot.emb.size != int(getattr(index, "dim", slot.emb.size)) or int(index.rows) <= 0:
            return
        epoch = index_epoch(index)
        # an index in a prefilter mode: the shared scan is the candidate scan too (the top-k of a query's 32 re-ranked candidates
        # is a prefix of their top-16, so what the inline search would return for k <= 16 is in the parked list, bit for bit)
        approx = bool(getattr(index, "prefilter", False)) and not hasattr(index, "ivf")
        k_pref = K_PREFETCH_APPROX if approx else K_PREFETCH
        batcher, cross = _batcher_for(st)
        with _lock:
            _waiting += 1
        try:
            if cross and not approx:
                scores, ids = await batcher.search(index, slot.emb, k_pref)
            else:
                if cross:       # the cross-index launch is the exact multi-index kernel: a prefilter index shares through its own batcher
                    if st.batcher is None:
                        from .batcher import QueryBatcher
                        st.batcher = QueryBatcher(st.index)
                    batcher = st.batcher
                scores, ids = await batcher.search(slot.emb, k_pref)
        finally:
            with _lock:
                _waiting -= 1
        slot.st, slot.epoch, slot.scores, slot.ids = st, epoch, scores, ids
        slot.k, slot.approx = k_pref, approx
        stats["prefetched"] += 1
    except asyncio.CancelledError:
        raise
    except Exception as e:          # the inline scan will answer (and report, if it fails too)
        stats["failed"] += 1
        logger.debug(f"k-NN prefetch skipped: {e}")


def _row_tag(st, row: int) -> Optional[int]:
    """The tag the row was written with (``docstore.IndexState.tag_of`` at add time), from its stored doc."""
    doc = st.row_doc[row] if 0 <= row < len(st.row_doc) else None
    if doc is None:
        return None
    pc = st.patients.lookup(doc.get("patientId"))
    dc = st.doc_types.lookup(doc.get("doc_type"))
    from .docstore import compose_tag
    return compose_tag(pc or 0, dc or 0)


def take(st, q: np.ndarray, k: int, fval: int, fmask: int) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """The synchronous search's part: (scores [k], ids [k]) exactly as the inline scan would return them, or None."""
    if config.RASS_KNN_PREFETCH <= 0:
        return None
    t = _task()
    slot = _slots.get(t) if t is not None else None
    if slot is None or slot.scores is None:
        return None
    if slot.st is not st or k > slot.k:
        return None
    if fmask and slot.approx:
        return None             # a filtered search of a prefilter index picks its candidates UNDER the filter: not derivable from this list
    qv = np.asarray(q, dtype=np.float32).reshape(-1)
    if qv.shape != slot.emb.shape or qv.tobytes() != slot.emb.tobytes():
        stats["other_query"] += 1
        return None
    if index_epoch(st.index) != slot.epoch:
        stats["stale"] += 1
        slot.scores = slot.ids = None
        return None
    scores, ids = slot.scores, slot.ids
    if not fmask:
        stats["answered"] += 1
        return scores[:k].copy(), ids[:k].copy()
    keep, exhausted = [], False
    with st.lock:
        for j in range(len(ids)):
            row = int(ids[j])
            if row < 0:             # fewer than 32 live rows: the list holds EVERY row, so the matches in it are all there are
                exhausted = True
                break
            tag = _row_tag(st, row)
            if tag is None:         # a row without a stored doc: its tag is not known on the host
                keep = None
                break
            if (tag & fmask) == fval:
                keep.append(j)
                if len(keep) == k:
                    break
    if keep is None or (len(keep) < k and not exhausted):
        stats["filter_short"] += 1
        return None
    stats["answered"] += 1
    out_s = np.full(k, -np.inf, dtype=np.float32)
    out_i = np.full(k, -1, dtype=np.int64)
    out_s[:len(keep)] = scores[keep]
    out_i[:len(keep)] = ids[keep]
    return out_s, out_i
