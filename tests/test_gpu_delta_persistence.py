"""Incremental persistence on the HIP index: IndexState.save_delta appends O(delta) segments (rows read back from HBM: the
stored bits), IndexState.load replays them on top of the snapshot through Engine.load_index + FlatIndex.add(normalize=False);
the restored index holds the same bits and answers like the live one."""
import asyncio
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_delta_segments_on_the_hip_index(gpu, tmp_path):
    from rassengine_amd import embedding, indexer
    from rassengine_amd.docstore import REGISTRY, IndexState
    from rassengine_amd.engine import Engine
    from tests.helpers import HashEmbedder
    eng = Engine(0, 1024)
    REGISTRY.clear()
    REGISTRY.set_index_factory(lambda name: eng.open_index(name))
    embedding.set_embedder(HashEmbedder(1024))
    try:
        name, prefix = "rass-idx-delta", str(tmp_path / "delta")
        docs = [{"doc_id": f"n-{i}", "doc_type": "unstructured", "patientId": f"p{i % 3}",
                 "unstructuredText": f"note {i} topic{i % 13} drug{i % 7}"} for i in range(900)]
        store = lambda u: asyncio.run(indexer.store_fhir_docs_in_opensearch([], u, None, name))
        store(docs[:500])
        st = REGISTRY.get(name)
        st.save(prefix)
        snap_bytes = os.path.getsize(os.path.join(str(tmp_path), [f for f in os.listdir(str(tmp_path)) if f.endswith(".rass")][0]))
        store(docs[500:700] + [dict(docs[7], unstructuredText="entirely new words")])
        assert st.save_delta(prefix)
        store(docs[700:] + [dict(docs[650], unstructuredText="other words")])
        assert st.save_delta(prefix)
        segs = sorted(f for f in os.listdir(str(tmp_path)) if f.endswith(".delta"))
        assert len(segs) == 2 and sum(os.path.getsize(os.path.join(str(tmp_path), f)) for f in segs) < snap_bytes
        st2 = IndexState.load("rass-idx-delta-restored", prefix, eng.load_index)
        assert st2.index.rows == st.index.rows == 902 and st2.index.count == st.index.count == 900
        assert st2.doc_row == st.doc_row and st2.row_doc == st.row_doc
        assert np.array_equal(st2.index.get_rows(0, 902), st.index.get_rows(0, 902))          # the stored bits
        for text, k, kw in (("note 650 topic0 drug6", 10, {}), ("entirely new words", 5, {}), ("note 3 topic3 drug3", 8, {"patient_id": "p0"})):
            q = asyncio.run(embedding.embed_query(text))
            a = indexer.HipIndexer(None, name).semantic_search(q, k=k, **kw)
            REGISTRY.put(st2)
            b = indexer.HipIndexer(None, "rass-idx-delta-restored").semantic_search(q, k=k, **kw)
            assert [d["doc_id"] for d, _ in a] == [d["doc_id"] for d, _ in b] and [s for _, s in a] == [s for _, s in b]
    finally:
        embedding.set_embedder(None)
        REGISTRY.set_index_factory(None)
        REGISTRY.clear()
        eng.close()
