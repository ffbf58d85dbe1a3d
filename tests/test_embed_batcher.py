"""CPU tests of the embed micro-batcher (rassengine_amd/batcher.EmbedBatcher behind embedding.py): concurrent
``embed_query`` / ``ollama_embed_text`` / ``embed_texts_in_batches`` coroutines (app/main.py:225-274; ``ask()``
awaits ``embed_query`` at app/main.py:2800, up to MAX_EMBED_CONCURRENCY requests in flight at 250-260) must meet in
ONE encoder call, keep the reference's order / dtype / blank handling, and keep its per-text error behaviour
(main.py raises, embedding_gen.py prints and returns zeros).  The encoder is a test double; the same scenarios run on
the HIP encoder in tests/test_gpu_embed_batcher.py."""
import asyncio
import threading
import time

import numpy as np
import pytest

from rassengine_amd import config, embedding
from rassengine_amd.batcher import EmbedBatcher
from tests.helpers import HashEmbedder


class SlowEmbedder(HashEmbedder):
    """HashEmbedder that takes a fixed time per call (like a forward) and fails on texts containing 'BAD'."""

    def __init__(self, dim=1024, seconds=0.005):
        super().__init__(dim)
        self.seconds = seconds
        self.threads = set()

    def encode(self, texts):
        self.threads.add(threading.get_ident())
        if any("BAD" in t for t in texts):
            self.calls.append(list(texts))
            raise RuntimeError("encoder refused a text")
        time.sleep(self.seconds)
        return super().encode(texts)


@pytest.fixture()
def emb():
    embedding.reset_batcher()
    e = SlowEmbedder()
    embedding.set_embedder(e)
    yield e
    embedding.reset_batcher()
    embedding.set_embedder(None)


def test_32_concurrent_queries_share_one_forward(emb):
    queries = [f"what is the blood pressure of patient {i}" for i in range(32)]

    async def go():
        return await asyncio.gather(*[embedding.embed_query(q) for q in queries])

    t0 = time.perf_counter()
    got = asyncio.run(go())
    wall = time.perf_counter() - t0
    assert len(emb.calls) <= 2, [len(c) for c in emb.calls]          # 32 requests, one or two encoder calls
    assert sorted(t for c in emb.calls for t in c) == sorted(queries)
    assert wall < 4 * emb.seconds + 0.05
    lone = HashEmbedder(1024)
    for q, g in zip(queries, got):
        assert g.shape == (1, config.EMBED_DIM) and g.dtype == np.float32 and g.flags["C_CONTIGUOUS"]
        assert np.array_equal(g[0], lone.encode([q])[0])                 # each caller got ITS row
    b = embedding.get_batcher()
    assert b.served == 32 and b.forwards == len(emb.calls)
    assert len(emb.threads) == 1                                         # one worker thread owns the encoder


def test_lone_caller_is_not_held_back(emb):
    emb.seconds = 0.0
    asyncio.run(embedding.embed_query("warm up the worker thread"))
    lat = []
    for i in range(20):
        t0 = time.perf_counter()
        asyncio.run(embedding.embed_query(f"single request {i}"))
        lat.append(time.perf_counter() - t0)
    # asyncio.run itself costs ~0.1-0.3 ms; the linger must not add more than max_delay on top
    assert sorted(lat)[len(lat) // 2] < 0.005, lat
    assert all(len(c) == 1 for c in emb.calls)


def test_mixed_callers_keep_order_blank_rows_and_shapes(emb):
    texts = ["alpha", "  ", "beta", "", "gamma"]

    async def go():
        return await asyncio.gather(embedding.embed_texts_in_batches(texts, batch_size=2),
                                    embedding.ollama_embed_text("delta"), embedding.embed_query("  "),
                                    embedding.embed_texts_in_batches([]), embedding.ollama_embed_text(" "),
                                    embedding.gen_embed_texts_in_batches(["epsilon", ""]),
                                    embedding.gen_embed_texts_in_batches([]))

    e, d, blank_q, empty, blank_t, g, g_empty = asyncio.run(go())
    ref = HashEmbedder(1024)
    assert e.shape == (5, 1024) and e.dtype == np.float32 and e.flags["C_CONTIGUOUS"]
    assert np.array_equal(e[[0, 2, 4]], ref.encode(["alpha", "beta", "gamma"]))
    assert not e[1].any() and not e[3].any()                  # blank -> zero row (app/main.py:227-228)
    assert isinstance(d, list) and len(d) == 1024 and np.array_equal(np.float32(d), ref.encode(["delta"])[0])
    assert blank_q.size == 0 and blank_q.shape == (0,)        # app/main.py:267-268
    assert empty.shape == (0,)                                # app/main.py:246-247
    assert blank_t == [0.0] * 1024
    assert g.shape == (2, 1024) and np.array_equal(g[0], ref.encode(["epsilon"])[0]) and not g[1].any()
    assert g_empty.shape == (0, 1024)                         # embedding_gen.py:174-175
    assert len(emb.calls) <= 2                                # everything non-blank met in one or two forwards
    assert all(t.strip() for c in emb.calls for t in c)       # blanks never reach the encoder


def test_a_failing_text_only_fails_its_own_request(emb, capsys):
    async def go():
        return await asyncio.gather(embedding.embed_query("good one"), embedding.embed_query("BAD text"),
                                    embedding.gen_ollama_embed_text("BAD again"), embedding.embed_query("good two"),
                                    embedding.gen_embed_texts_in_batches(["fine", "BAD inside", "also fine"]),
                                    return_exceptions=True)

    a, b, c, d, g = asyncio.run(go())
    ref = HashEmbedder(1024)
    assert np.array_equal(a[0], ref.encode(["good one"])[0]) and np.array_equal(d[0], ref.encode(["good two"])[0])
    assert isinstance(b, RuntimeError)                        # main.py flavour raises (raise_for_status, 235)
    assert c == [0.0] * 1024                                  # embedding_gen flavour: printed, zero vector (168-170)
    assert np.array_equal(g[[0, 2]], ref.encode(["fine", "also fine"])) and not g[1].any()
    assert "[ERROR] Ollama embed request" in capsys.readouterr().out
    assert embedding.get_batcher().retries >= 2               # the coalesced forward failed, entries re-ran alone


def test_upload_slices_run_alone_and_queries_overtake_them(emb):
    emb.seconds = 0.02
    big = [f"chunk {i}" for i in range(3 * embedding.UPLOAD_SLICE)]

    async def go():
        up = asyncio.ensure_future(embedding.embed_texts_in_batches(big))
        await asyncio.sleep(0.005)                            # the first slice is in its forward now
        t0 = time.perf_counter()
        q = await embedding.embed_query("a query during the upload")
        return await up, q, time.perf_counter() - t0

    e, q, q_wait = asyncio.run(go())
    ref = HashEmbedder(1024)
    assert e.shape == (len(big), 1024) and np.array_equal(e[[0, -1]], ref.encode([big[0], big[-1]]))
    sizes = [len(c) for c in emb.calls]
    assert sizes.count(embedding.UPLOAD_SLICE) == 3 and sizes.count(1) == 1
    assert sizes.index(1) <= 1                                # the query went right after the slice in flight
    assert np.array_equal(q[0], ref.encode(["a query during the upload"])[0])
    assert q_wait > 0                                         # (its wait = the rest of ONE slice, checked by the order)


def test_two_event_loops_share_the_batcher(emb):
    emb.seconds = 0.01
    out = {}

    def client(name, n):
        async def go():
            return await asyncio.gather(*[embedding.embed_query(f"{name} {i}") for i in range(n)])
        out[name] = asyncio.run(go())

    ts = [threading.Thread(target=client, args=(f"loop{j}", 8)) for j in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    ref = HashEmbedder(1024)
    for name, rows in out.items():
        for i, r in enumerate(rows):
            assert np.array_equal(r[0], ref.encode([f"{name} {i}"])[0])
    assert len(emb.calls) < 24 and len(emb.threads) == 1


def test_cancelled_caller_does_not_break_the_batch(emb):
    emb.seconds = 0.02

    async def go():
        a = asyncio.ensure_future(embedding.embed_query("stays"))
        b = asyncio.ensure_future(embedding.embed_query("leaves"))
        await asyncio.sleep(0.002)
        b.cancel()
        return await a

    a = asyncio.run(go())
    assert np.array_equal(a[0], HashEmbedder(1024).encode(["stays"])[0])
    # the batcher still serves
    assert asyncio.run(embedding.embed_query("next")).shape == (1, 1024)


def test_coalescing_can_be_switched_off(emb, monkeypatch):
    monkeypatch.setattr(config, "RASS_EMBED_BATCH_MAX", 0)

    async def go():
        return await asyncio.gather(*[embedding.embed_query(f"q{i}") for i in range(6)])

    got = asyncio.run(go())
    assert embedding.get_batcher() is None and len(emb.calls) == 6 and all(g.shape == (1, 1024) for g in got)


def test_sustained_queries_cannot_starve_an_upload_slice():
    """small entries go first, but a waiting upload slice is served after at most ``big_after`` small
    batches in a row."""
    order = []

    def enc(texts):
        order.append(len(texts))
        time.sleep(0.002)
        return np.zeros((len(texts), 8), dtype=np.float32)

    b = EmbedBatcher(enc, max_seqs=1, max_delay_ms=0.0)        # one query per forward: 40 queued queries = 40 small batches

    async def go():
        qs = [asyncio.ensure_future(b.embed([f"q{i}"])) for i in range(40)]
        up = asyncio.ensure_future(b.embed([f"chunk{i}" for i in range(5)]))     # > max_seqs: an upload slice
        await asyncio.gather(up, *qs)

    asyncio.run(go())
    b.close()
    assert sorted(order) == [1] * 40 + [5]
    assert order.index(5) <= b.big_after + 1, order            # not behind all 40 queries
    assert order.index(5) >= 1                                 # but queries do go first


def test_batcher_caps_and_close():
    calls = []

    def enc(texts):
        calls.append(len(texts))
        time.sleep(0.005)
        return np.zeros((len(texts), 8), dtype=np.float32)

    b = EmbedBatcher(enc, max_seqs=8, max_delay_ms=0.2)

    async def go():
        return await asyncio.gather(*[b.embed([f"t{i}", f"u{i}"]) for i in range(10)])

    rows = asyncio.run(go())
    assert all(r.shape == (2, 8) for r in rows)
    assert max(calls) <= 8 and sum(calls) == 20               # never more than max_seqs sequences per forward
    b.close()
    with pytest.raises(RuntimeError):
        asyncio.run(b.embed(["late"]))
    with pytest.raises(ValueError):
        EmbedBatcher(enc, max_seqs=0)


# ------------------------------------------------------------------ CrossIndexBatcher: who may share a launch
class _FakeIndex:
    """FlatIndex's surface as far as the batcher sees it: search() + multi_tiles."""

    def __init__(self, name, rows, dtype="f32", gids=False):
        self.name, self.rows, self.dtype, self.has_global_ids = name, rows, dtype, gids
        self.solo_calls = []

    @property
    def multi_tiles(self):
        if self.dtype != "f32" or self.has_global_ids:
            return 0
        t = max(1, (self.rows + 31) // 32)
        return t if t <= 65536 // 2 else 0

    def search(self, qs, k, f=None, m=None):
        self.solo_calls.append(qs.shape[0])
        return (np.full((qs.shape[0], k), 0.5, np.float32), np.full((qs.shape[0], k), hash(self.name) % 1000, np.int64))


class _FakeEngine:
    """rass_index_search_multi's admission rules (api.hip): fp32, plain ids, <= 65 536 tiles over the distinct indices."""

    def __init__(self):
        self.calls = []

    def search_multi(self, indices, qs, k, f=None, m=None):
        tiles = sum((ix.rows + 31) // 32 for ix in {id(i): i for i in indices}.values())
        if tiles > 65536 or any(ix.dtype != "f32" or ix.has_global_ids for ix in indices):
            raise RuntimeError("RASS_ERR_UNSUPPORTED")
        self.calls.append([ix.name for ix in indices])
        return (np.full((len(indices), k), 0.25, np.float32),
                np.stack([np.full(k, hash(ix.name) % 1000, np.int64) for ix in indices]))


def test_cross_index_batcher_never_fails_a_batch_over_its_company():
    """One user with 3 M rows, two users with 1.1 M rows each (together over the 2 M-row budget), a bf16 index and a
    shard with global ids in the SAME 20 ms window as small per-user indices: everybody is answered."""
    from rassengine_amd.batcher import CrossIndexBatcher
    eng = _FakeEngine()
    small = [_FakeIndex(f"small{i}", 10_000) for i in range(6)]
    huge = _FakeIndex("huge", 3_000_000)
    mid = [_FakeIndex("midA", 1_040_000), _FakeIndex("midB", 1_040_000)]   # each under half the budget, together over it
    bf16 = _FakeIndex("bf16", 50_000, dtype="bf16")
    shard = _FakeIndex("shard", 50_000, gids=True)
    b = CrossIndexBatcher(eng, max_batch=32, max_delay_ms=20.0)
    who = small + [huge, huge, mid[0], mid[1], bf16, shard, small[0]]

    async def go():
        out = await asyncio.gather(*[b.search(ix, np.ones(8, np.float32), 3) for ix in who])
        await b.close()
        return out

    out = asyncio.run(go())
    assert len(out) == len(who)
    for ix, (s, i) in zip(who, out):
        assert s.shape == (3,) and int(i[0]) == hash(ix.name) % 1000      # everyone got an answer from ITS index
    assert huge.solo_calls == [2]                   # too large for any cross-index batch: ONE scan of its own for both
    assert bf16.solo_calls == [1] and shard.solo_calls == [1]
    flat = [n for c in eng.calls for n in c]
    assert sorted(flat) == sorted([ix.name for ix in small] + ["small0", "midA", "midB"])
    assert len(eng.calls) == 2                      # the two mid-size users could not share ONE launch: two groups
    assert not ({"midA", "midB"} <= set(eng.calls[0])) and not ({"midA", "midB"} <= set(eng.calls[1]))
    assert b.served == len(who)


def test_cross_index_batcher_falls_back_when_the_engine_refuses():
    from rassengine_amd.batcher import CrossIndexBatcher

    class Refusing(_FakeEngine):
        def search_multi(self, *a, **k):
            raise RuntimeError("RASS_ERR_UNSUPPORTED: grew past the budget since it was planned")

    a, c = _FakeIndex("a", 1000), _FakeIndex("c", 2000)
    b = CrossIndexBatcher(Refusing(), max_delay_ms=20.0)

    async def go():
        out = await asyncio.gather(b.search(a, np.ones(8, np.float32), 2), b.search(c, np.ones(8, np.float32), 2),
                                   b.search(a, np.ones(8, np.float32), 1))
        await b.close()
        return out

    out = asyncio.run(go())
    assert [o[0].shape for o in out] == [(2,), (2,), (1,)] and a.solo_calls == [2] and c.solo_calls == [1]
