"""CPU tests of the k-NN prefetch (rassengine_amd/prefetch.py): ``ask()``-shaped coroutines
(tests/fake_reference.py: embed_query -> ensure_index_exists -> OpenSearchIndexer(...) -> a SYNCHRONOUS search,
app/main.py:2800-2885) through the installed shim, on the oracle-backed index double.  Concurrent requests must share
scans at the ``await ensure_index_exists`` and every answer must be the serial path's answer, exactly."""
import asyncio

import numpy as np
import pytest

from rassengine_amd import config, embedding, indexer, prefetch
from rassengine_amd.docstore import REGISTRY
from tests import fake_reference
from tests.helpers import HashEmbedder, OracleIndex


class CountingIndex(OracleIndex):
    def __init__(self, dim):
        super().__init__(dim)
        self.calls = []

    def search(self, queries, k, q_filter=None, q_filter_mask=None):
        self.calls.append((queries.shape[0], k, q_filter is not None))
        return super().search(queries, k, q_filter, q_filter_mask)


def _docs(n):
    docs = [{"doc_id": f"d{i}", "doc_type": "unstructured", "patientId": f"p{i % 4}",
             "unstructuredText": f"note {i} topic{i % 9} drug{i % 5}"} for i in range(n)]
    # duplicated rows: identical vectors -> ties, broken by row id
    for i in range(0, n, 10):
        docs[i]["unstructuredText"] = "the very same words"
    # one rare patient: a filter on it finds < k matches in any top-32
    docs[3]["patientId"] = "rare"
    return docs


@pytest.fixture
def world():
    REGISTRY.clear()
    fake_reference.reset()
    prefetch.reset_stats()
    idx = CountingIndex(1024)
    REGISTRY.set_index_factory(lambda name: idx)
    embedding.set_embedder(HashEmbedder(1024))
    embedding.reset_batcher()
    m = fake_reference.make_module()
    indexer.install(m)
    mode0 = config.RASS_KNN_PREFETCH
    asyncio.run(m.store_fhir_docs_in_opensearch([], _docs(200), None, "idx-p"))
    yield m, idx
    config.RASS_KNN_PREFETCH = mode0
    indexer.uninstall(m)
    embedding.reset_batcher()
    embedding.set_embedder(None)
    REGISTRY.set_index_factory(None)
    REGISTRY.clear()


def _plain(hits):
    return [(d["doc_id"], float(s)) for d, s in hits]


REQUESTS = ([("SEMANTIC", f"note {i} topic{i % 9} drug{i % 5}", 5, None) for i in range(20)]
            + [("SEMANTIC", "the very same words", 7, None)] * 2                   # ties
            + [("HYBRID", "note 5 topic5 drug0", 3, None), ("MULTI_INTENT", "topic3 drug3", 10, None),
               ("SEMANTIC", "note 8 topic8", 5, "p0"), ("HYBRID", "note 9 topic0", 4, "p1"),
               ("SEMANTIC", "note 3 topic3 drug3", 5, "rare"),                     # 1 match: list not exhausted -> fallback
               ("SEMANTIC", "drug2", 32, None), ("SEMANTIC", "drug1", 40, None),   # k = 32 served, k > 32 falls back
               ("HYBRID_STRUCTURED", "note 1", 3, None),                           # doc_type never indexed -> []
               ("KEYWORD", "note 2", 3, None), ("SEMANTIC", "   ", 3, None)])      # text engine / blank query


async def _all(m, reqs, client=None):
    return await asyncio.gather(*[m.ask_shaped(q, intent, k, client, "idx-p", primary_patient_id=pid)
                                  for intent, q, k, pid in reqs])


def test_concurrent_ask_requests_share_scans_and_equal_the_serial_path(world):
    m, idx = world
    config.RASS_KNN_PREFETCH = 0
    idx.calls.clear()
    serial = asyncio.run(_all(m, REQUESTS))
    serial_scans = len(idx.calls)
    config.RASS_KNN_PREFETCH = 1
    idx.calls.clear()
    prefetch.reset_stats()
    got = asyncio.run(_all(m, REQUESTS))
    assert len(got) == len(serial)
    for (intent, q, k, pid), a, b in zip(REQUESTS, got, serial):
        assert _plain(a) == _plain(b), (intent, q, k, pid)        # same docs, same order, same float scores
    knn = sum(1 for r in REQUESTS if r[0] in ("SEMANTIC", "HYBRID", "MULTI_INTENT") and r[1].strip())
    assert serial_scans == knn
    # 29 embeddings -> prefetched in <= 2 shared scans; the three fallbacks (rare patient, k = 40) scan inline
    shared = [c for c in idx.calls if c[1] == prefetch.K_PREFETCH and c[0] > 1]
    assert 1 <= len(shared) <= 2 and sum(c[0] for c in shared) >= knn - 2
    assert len(idx.calls) <= 2 + 3
    assert prefetch.stats["answered"] >= knn - 3 and prefetch.stats["filter_short"] == 1
    ties = got[20]
    assert [d for d, _ in _plain(ties)][:3] == ["d0", "d10", "d20"]      # equal scores: row id ascending


def test_lone_request_scans_inline(world):
    m, idx = world
    config.RASS_KNN_PREFETCH = 1
    idx.calls.clear()
    hits = asyncio.run(m.ask_shaped("note 7 topic7 drug2", "SEMANTIC", 3, None, "idx-p"))
    assert hits[0][0]["doc_id"] == "d7"
    assert idx.calls == [(1, 3, False)]                   # today's inline scan: k as asked, nothing prefetched
    assert prefetch.stats["alone"] == 1 and prefetch.stats["prefetched"] == 0
    # a KEYWORD request alone: no scan at all
    idx.calls.clear()
    asyncio.run(m.ask_shaped("note 7", "KEYWORD", 3, fake_reference.FakeClient(), "idx-p"))
    assert idx.calls == []


def test_write_between_prefetch_and_use_falls_back(world):
    m, idx = world
    config.RASS_KNN_PREFETCH = 2                          # always: a single coroutine is enough

    async def racing(write):
        emb = await m.embed_query("note 42 topic6 drug2")
        await m.ensure_index_exists(None, "idx-p")        # the top-32 is parked on this task now
        await write()
        return m.OpenSearchIndexer(None, "idx-p").semantic_search(query_emb=emb, k=4, query="x")

    async def nothing():
        pass

    async def overwrite():                                # tombstones d42's row, appends a new one
        await m.store_fhir_docs_in_opensearch([], [{"doc_id": "d42", "doc_type": "unstructured", "patientId": "p2",
                                                     "unstructuredText": "something else entirely"}], None, "idx-p")

    prefetch.reset_stats()
    before = asyncio.run(racing(nothing))
    assert before[0][0]["doc_id"] == "d42" and prefetch.stats["answered"] == 1
    after = asyncio.run(racing(overwrite))
    assert prefetch.stats["stale"] == 1
    assert "d42" not in [d["doc_id"] for d, _ in after] or after[0][0]["doc_id"] != "d42"
    config.RASS_KNN_PREFETCH = 0
    ref = asyncio.run(racing(nothing))
    assert _plain(after) == _plain(ref)


def test_filters_answered_from_the_parked_list_only_when_exact(world):
    m, idx = world
    config.RASS_KNN_PREFETCH = 2
    ix = m.OpenSearchIndexer(None, "idx-p")

    async def one(q, k, **kw):
        emb = await m.embed_query(q)
        await m.ensure_index_exists(None, "idx-p")
        return m.OpenSearchIndexer(None, "idx-p").semantic_search(query_emb=emb, k=k, **kw), emb

    for q, k, kw in (("note 8 topic8 drug3", 3, {"patient_id": "p0"}),
                     ("topic4", 6, {"filter_clause": {"term": {"patientId": "p1"}}}),
                     ("drug0", 2, {"filter_clause": {"term": {"doc_type": "unstructured"}}, "patient_id": "p3"}),
                     ("note 3 topic3 drug3", 2, {"patient_id": "rare"}),           # 1 match in the index
                     ("drug4", 5, {"patient_id": "nobody"}),                       # never indexed -> []
                     ("drug4", 5, {"filter_clause": ["ner", "entities"]})):        # quirk 1: ignored
        prefetch.reset_stats()
        idx.calls.clear()
        got, emb = asyncio.run(one(q, k, **kw))
        scans = len(idx.calls)
        config.RASS_KNN_PREFETCH = 0
        ref = ix.semantic_search(emb, k=k, **kw)
        config.RASS_KNN_PREFETCH = 2
        assert _plain(got) == _plain(ref), (q, kw)
        if kw.get("patient_id") == "rare":
            assert prefetch.stats["filter_short"] == 1 and scans == 2
        elif kw.get("patient_id") == "nobody":
            assert got == []
        else:
            assert prefetch.stats["answered"] == 1 and scans == 1, (q, kw, prefetch.stats)


def test_small_index_filter_is_answered_exactly_from_an_exhausted_list(world):
    """Fewer than 32 live rows: the parked list holds every row, so ANY filter is answered from it."""
    m, idx = world
    config.RASS_KNN_PREFETCH = 2
    docs = [{"doc_id": f"s{i}", "doc_type": "unstructured", "patientId": f"p{i % 3}",
             "unstructuredText": f"small {i}"} for i in range(9)]
    small = CountingIndex(1024)
    REGISTRY.set_index_factory(lambda name: small)
    asyncio.run(m.store_fhir_docs_in_opensearch([], docs, None, "idx-small"))

    async def one():
        emb = await m.embed_query("small 4")
        await m.ensure_index_exists(None, "idx-small")
        return m.OpenSearchIndexer(None, "idx-small").semantic_search(query_emb=emb, k=5, patient_id="p1"), emb

    small.calls.clear()
    got, emb = asyncio.run(one())
    assert len(small.calls) == 1 and prefetch.stats["answered"] >= 1
    config.RASS_KNN_PREFETCH = 0
    ref = m.OpenSearchIndexer(None, "idx-small").semantic_search(query_emb=emb, k=5, patient_id="p1")
    assert _plain(got) == _plain(ref) and len(got) == 3 and got[0][0]["doc_id"] == "s4"


def test_off_switch_and_foreign_embeddings(world):
    m, idx = world
    config.RASS_KNN_PREFETCH = 2

    async def other_vector():
        await m.embed_query("note 1 topic1 drug1")
        await m.ensure_index_exists(None, "idx-p")
        emb2 = np.asarray(embedding.get_embedder().encode(["note 2 topic2 drug2"]), dtype=np.float32)
        return m.OpenSearchIndexer(None, "idx-p").semantic_search(query_emb=emb2, k=3)

    prefetch.reset_stats()
    hits = asyncio.run(other_vector())
    assert hits[0][0]["doc_id"] == "d2" and prefetch.stats["other_query"] == 1 and prefetch.stats["answered"] == 0
    # a synchronous caller outside any event loop: no task, no prefetch, same answer
    emb = asyncio.run(m.embed_query("note 1 topic1 drug1"))
    assert m.OpenSearchIndexer(None, "idx-p").semantic_search(emb, k=1)[0][0]["doc_id"] == "d1"
    config.RASS_KNN_PREFETCH = 0
    idx.calls.clear()
    asyncio.run(_all(m, REQUESTS[:8]))
    assert len(idx.calls) == 8 and all(c[0] == 1 for c in idx.calls)
