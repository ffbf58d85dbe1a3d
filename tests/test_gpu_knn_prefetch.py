"""The k-NN prefetch (rassengine_amd/prefetch.py) over the REAL HIP index: 32 concurrent ask()-shaped
coroutines (tests/fake_reference.py; app/main.py:2800-2885) on a 1 M-row index share <= 2 scan launches at their
``await ensure_index_exists`` and every answer is bit-identical to the serial path's (prefetch off)."""
import asyncio

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _plain(hits):
    return [(d["doc_id"], float(s)) for d, s in hits]


@pytest.fixture
def served(gpu):
    from rassengine_amd import config, embedding, indexer, prefetch
    from rassengine_amd.docstore import REGISTRY
    from rassengine_amd.engine import Engine
    from tests import fake_reference as FR
    from tests.helpers import HashEmbedder

    eng = Engine(0, 1024)
    REGISTRY.clear()
    REGISTRY.set_index_factory(lambda name: eng.open_index(name))
    embedding.set_embedder(HashEmbedder(1024))
    embedding.reset_batcher()
    FR.reset()
    main = FR.make_module("main")
    indexer.install(main)
    mode0 = config.RASS_KNN_PREFETCH
    prefetch.reset_stats()
    yield main, eng
    config.RASS_KNN_PREFETCH = mode0
    embedding.reset_batcher()
    embedding.set_embedder(None)
    REGISTRY.set_index_factory(None)
    REGISTRY.clear()
    indexer._ORIGINALS.clear()
    eng.close()


def _small_docs(n=3000):
    docs = [{"doc_id": f"d{i}", "doc_type": "unstructured", "patientId": f"p{i % 4}",
             "unstructuredText": f"note {i} topic{i % 9} drug{i % 5}"} for i in range(n)]
    for i in range(0, n, 500):
        docs[i]["unstructuredText"] = "the very same words"          # identical rows: ties by row id
    docs[3]["patientId"] = "rare"
    return docs


def test_32_concurrent_asks_on_1M_rows_share_two_launches(served):
    from rassengine_amd import config, prefetch
    from rassengine_amd.docstore import REGISTRY
    main, eng = served
    n = 1_000_000
    st = REGISTRY.get("idx-big", create=True)
    st.index.fill_synthetic(n, seed=7)
    st.row_doc = [{"doc_id": f"doc-{i}"} for i in range(n)]
    st.doc_row = {}
    eng.synchronize()
    texts = [f"note {i} topic{i % 9} drug{i % 5}" for i in range(32)]

    async def burst():
        return await asyncio.gather(*[main.ask_shaped(t, "SEMANTIC", 5, None, "idx-big") for t in texts])

    config.RASS_KNN_PREFETCH = 0
    eng.kernel_timing_begin(128)
    serial = asyncio.run(burst())
    _, serial_launches = eng.kernel_timing_end()
    assert serial_launches == 32
    config.RASS_KNN_PREFETCH = 1
    asyncio.run(burst())                                             # warm the batcher thread / executor
    prefetch.reset_stats()
    eng.kernel_timing_begin(128)
    got = asyncio.run(burst())
    ms, launches = eng.kernel_timing_end()
    assert launches <= 2, launches                                   # 32 requests, <= 2 HBM passes
    assert prefetch.stats["answered"] == 32
    for a, b in zip(got, serial):
        assert len(a) == 5 and _plain(a) == _plain(b)                # same rows, same order, same float scores
    print(f"32 ask-shaped requests on {n} rows: {launches} scan launch(es), {ms:.3f} ms of scan kernels "
          f"(serial: {serial_launches} launches)")


def test_prefetch_equals_serial_with_ties_filters_writes_and_two_indices(served):
    from rassengine_amd import config, prefetch
    from rassengine_amd.docstore import REGISTRY
    main, eng = served
    docs = _small_docs()
    asyncio.run(main.store_fhir_docs_in_opensearch([], docs, None, "idx-a"))
    other = [dict(d, doc_id="o" + d["doc_id"]) for d in docs[:700]]
    asyncio.run(main.store_fhir_docs_in_opensearch([], other, None, "idx-b"))
    reqs = ([("SEMANTIC", f"note {i} topic{i % 9} drug{i % 5}", 5, None, "idx-a" if i % 3 else "idx-b") for i in range(20)]
            + [("SEMANTIC", "the very same words", 7, None, "idx-a"), ("HYBRID", "note 5 topic5 drug0", 3, None, "idx-a"),
               ("MULTI_INTENT", "topic3 drug3", 10, None, "idx-b"), ("SEMANTIC", "note 8 topic8", 5, "p0", "idx-a"),
               ("HYBRID", "note 9 topic0", 4, "p1", "idx-b"), ("SEMANTIC", "note 3 topic3 drug3", 5, "rare", "idx-a"),
               ("SEMANTIC", "drug2", 32, None, "idx-a"), ("SEMANTIC", "drug1", 40, None, "idx-a"),
               ("HYBRID_STRUCTURED", "note 1", 3, None, "idx-a"), ("SEMANTIC", "  ", 3, None, "idx-a")])

    async def burst():
        return await asyncio.gather(*[main.ask_shaped(q, intent, k, None, name, primary_patient_id=pid)
                                      for intent, q, k, pid, name in reqs])

    config.RASS_KNN_PREFETCH = 0
    serial = asyncio.run(burst())
    config.RASS_KNN_PREFETCH = 1
    prefetch.reset_stats()
    eng.kernel_timing_begin(128)
    got = asyncio.run(burst())
    _, launches = eng.kernel_timing_end()
    for r, a, b in zip(reqs, got, serial):
        assert _plain(a) == _plain(b), r
    assert [d for d, _ in _plain(got[20])][:3] == ["d0", "d500", "d1000"]
    knn = sum(1 for r in reqs if r[0] in ("SEMANTIC", "HYBRID", "MULTI_INTENT") and r[1].strip())
    assert prefetch.stats["answered"] >= knn - 3 and prefetch.stats["filter_short"] == 1
    assert launches <= 2 + 3 + 2, launches          # shared launches + the fallbacks (rare: 1, k = 40: 2 passes)

    # a write landing between the prefetch and its use
    config.RASS_KNN_PREFETCH = 2

    async def racing(write):
        emb = await main.embed_query("note 42 topic6 drug2")
        await main.ensure_index_exists(None, "idx-a")
        if write:
            await main.store_fhir_docs_in_opensearch([], [dict(docs[42], unstructuredText="something else entirely")],
                                                     None, "idx-a")
        return main.OpenSearchIndexer(None, "idx-a").semantic_search(query_emb=emb, k=4, query="x")

    prefetch.reset_stats()
    before = asyncio.run(racing(False))
    assert before[0][0]["doc_id"] == "d42" and prefetch.stats["answered"] == 1
    after = asyncio.run(racing(True))
    assert prefetch.stats["stale"] == 1 and after[0][0]["doc_id"] != "d42"
    config.RASS_KNN_PREFETCH = 0
    assert _plain(after) == _plain(asyncio.run(racing(False)))
    st = REGISTRY.get("idx-a")
    assert st.index.rows == len(docs) + 1 and st.index.count == len(docs)
