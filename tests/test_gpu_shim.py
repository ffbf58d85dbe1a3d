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
