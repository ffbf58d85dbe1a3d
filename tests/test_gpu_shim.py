"""The reference-shaped shim (HipIndexer / store_fhir_docs_in_opensearch) over the REAL HIP
index: an unmodified ask()-shaped caller gets the oracle's ranking (SURVEY §4 iv)."""
import asyncio

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ask_shaped_flow_on_hip_index(gpu, oracle):
    from rassengine_amd import config, embedding, indexer
    from rassengine_amd.docstore import REGISTRY
    from rassengine_amd.engine import Engine
    from tests.helpers import HashEmbedder

    eng = Engine(0, 1024)
    REGISTRY.clear()
    REGISTRY.set_index_factory(lambda name: eng.open_index(name))
    emb = HashEmbedder(1024)
    embedding.set_embedder(emb)
    try:
        name = "rass-idx-user9"
        docs = [{"doc_id": f"text-f-{i}", "doc_type": "unstructured", "patientId": f"p{i % 3}",
                 "unstructuredText": f"note {i} mentions condition{i % 11} and drug{i % 5}"} for i in range(300)]
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs, None, name))

        # --- what ask() does (app/main.py:2800-2802, 2878-2885)
        query = "note about condition7 and drug2"
        query_emb = asyncio.run(embedding.embed_query(query))
        asyncio.run(indexer.ensure_index_exists(None, name))
        os_indexer = indexer.HipIndexer(None, name)
        assert os_indexer.has_any_data()
        partial = os_indexer.semantic_search(query=query, query_emb=query_emb, k=5, filter_clause=None,
                                             patient_id="p1")
        assert len(partial) == 5 and all(d["patientId"] == "p1" for d, _ in partial)

        texts = [d["unstructuredText"] for d in docs]
        xn = oracle.normalize_ref(asyncio.run(embedding.embed_texts_in_batches(texts))).astype(np.float32)
        tags = np.array([i % 3 + 1 for i in range(300)], dtype=np.int32)  # dictionary codes p0->1, p1->2, p2->3
        rs, ri = oracle.search(xn, oracle.normalize_ref(query_emb).astype(np.float32), 5, tags=tags,
                               qfilter=np.array([2], dtype=np.int32))
        assert [d["doc_id"] for d, _ in partial] == [docs[i]["doc_id"] for i in ri[0]]
        assert np.allclose([s for _, s in partial], 1.0 / (2.0 - rs[0]), atol=1e-5)

        # overwrite through the shim tombstones the old row on the GPU
        asyncio.run(indexer.store_fhir_docs_in_opensearch(
            [], [dict(docs[7], unstructuredText="entirely new content here")], None, name))
        q2 = asyncio.run(embedding.embed_query(docs[7]["unstructuredText"]))
        ids = [d["doc_id"] for d, _ in os_indexer.semantic_search(q2, k=10)]
        assert ids.count("text-f-7") <= 1
        st = REGISTRY.get(name)
        assert st.index.count == 300 and st.index.rows == 301
    finally:
        embedding.set_embedder(None)
        REGISTRY.set_index_factory(None)
        REGISTRY.clear()
        eng.close()


def test_every_intent_of_ask_on_the_hip_index(gpu, oracle):
    """install() on a module with the reference's surface (tests/fake_reference.py): the 11-entry method table
    and both call shapes of ask() (app/main.py:2855-2892) work for every intent with the HIP index behind the
    four knn-bearing builders and the module's own OpenSearchIndexer behind the other eight; k > 32 passes
    through; the doc_type filter of hybrid_structured_search runs as a masked tag compare on the GPU."""
    from rassengine_amd import embedding, indexer
    from rassengine_amd.docstore import REGISTRY
    from rassengine_amd.engine import Engine
    from tests import fake_reference as FR
    from tests.helpers import HashEmbedder
    from tests.test_dropin_boundary import ALL_INTENTS, exercise_all_intents

    eng = Engine(0, 1024)
    REGISTRY.clear()
    REGISTRY.set_index_factory(lambda name: eng.open_index(name))
    embedding.set_embedder(HashEmbedder(1024))
    FR.reset()
    main = FR.make_module("main")
    indexer.install(main)
    try:
        name = "rass-idx-user12"
        docs = [{"doc_id": f"text-f-{i}", "doc_type": "unstructured", "patientId": f"p{i % 3}",
                 "unstructuredText": f"chunk number {i} about topic{i % 7}"} for i in range(200)]
        twin = {"doc_id": "Condition-9", "doc_type": "structured", "patientId": "p0",
                "unstructuredText": "chunk number 3 about topic3 structured twin"}
        client = FR.FakeClient([(docs[3], 7.5), (docs[10], 4.0)])
        asyncio.run(main.store_fhir_docs_in_opensearch([], docs + [twin], client, name))
        assert len(FR.BULKED) == 201 and REGISTRY.get(name).index.count == 201
        out = exercise_all_intents(main, client, name)
        assert set(out) == set(ALL_INTENTS)
        assert {c[0] for c in FR.CALLS} >= set(FR.TEXT_METHODS)
        sem = out["SEMANTIC"]
        assert len(sem) == 5 and all(d["patientId"] == "p0" for d, _ in sem) and sem[0][0]["doc_id"] == "text-f-3"
        assert out["HYBRID"][0][0]["doc_id"] == "text-f-3" and abs(out["HYBRID"][0][1] - 9.5) < 1e-4
        assert abs(out["MULTI_INTENT"][0][1] - 9.0) < 1e-4
        hs = dict((d["doc_id"], s) for d, s in out["HYBRID_STRUCTURED"])
        assert "Condition-9" in hs and 1.0 < hs["Condition-9"] <= 2.0           # knn x 2.0 on the structured row only
        assert hs["text-f-3"] == 7.5                                           # text clause only: filtered out of the knn
        # the same rows the oracle ranks, through the masked filter (patient p0 AND doc_type structured)
        ix = main.OpenSearchIndexer(None, name)
        q = asyncio.run(main.embed_query("chunk number 3 about topic3"))
        assert [d["doc_id"] for d, _ in ix.hybrid_structured_search("x", q, k=5, patient_id="p0")] == ["Condition-9"]
        assert ix.hybrid_structured_search("x", q, k=5, patient_id="p1") == []
        big = ix.semantic_search(q, k=150)                                      # k > 32: passes on the GPU, no clamp
        assert len(big) == 150 and len({d["doc_id"] for d, _ in big}) == 150
        texts = [d["unstructuredText"] for d in docs + [twin]]
        xn = oracle.normalize_ref(asyncio.run(embedding.embed_texts_in_batches(texts))).astype(np.float32)
        rs, ri = oracle.search(xn, oracle.normalize_ref(q).astype(np.float32), 150)
        # the oracle's order; two rows may only trade places if their scores are a rounding apart (the GPU's normalise is
        # within 2 ulp of numpy's).  The texts' vectors come from crc32-seeded words: the same in every process.
        want = [(docs + [twin])[i]["doc_id"] for i in ri[0]]
        score_of = dict(zip(want, rs[0]))
        for got_doc, want_doc in zip([d["doc_id"] for d, _ in big], want):
            assert got_doc == want_doc or abs(score_of[got_doc] - score_of[want_doc]) <= 4e-6, (got_doc, want_doc)
    finally:
        embedding.set_embedder(None)
        REGISTRY.set_index_factory(None)
        REGISTRY.clear()
        indexer._ORIGINALS.clear()
        eng.close()
