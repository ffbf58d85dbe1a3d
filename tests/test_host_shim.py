"""CPU tests of the host-side mirror of the reference boundary (rassengine_amd/indexer.py,
embedding.py, docstore.py): same names, argument meaning and error behaviour as
app/main.py:225-274, 1211-1282, 1395-1560.  The arithmetic behind the index is a test double
(tests/helpers.OracleIndex); the GPU tests run the same shim on the HIP index."""
import asyncio
import os

import numpy as np
import pytest

os.environ.setdefault("PYTHONHASHSEED", "0")

from rassengine_amd import config, embedding, indexer  # noqa: E402
from rassengine_amd.docstore import REGISTRY  # noqa: E402
from tests.helpers import HashEmbedder, OracleIndex  # noqa: E402


@pytest.fixture()
def shim():
    REGISTRY.clear()
    REGISTRY.set_index_factory(lambda name: OracleIndex(1024))
    emb = HashEmbedder(1024)
    embedding.set_embedder(emb)
    yield emb
    embedding.set_embedder(None)
    REGISTRY.set_index_factory(None)
    REGISTRY.clear()


def run(coro):
    return asyncio.run(coro)


def _docs(n, patient=None, prefix="text-note"):
    return [{"doc_id": f"{prefix}-{i}", "doc_type": "unstructured", "resourceType": None, "file_path": "/x",
             "file_type": "text", "patientId": patient, "unstructuredText": f"chunk number {i} about topic{i % 7}"}
            for i in range(n)]


def test_embed_contract_matches_reference(shim):
    # a1: blank -> zeros list of EMBED_DIM (app/main.py:227-228)
    assert run(embedding.ollama_embed_text("   ")) == [0.0] * config.EMBED_DIM
    v = run(embedding.ollama_embed_text("diabetes mellitus"))
    assert isinstance(v, list) and len(v) == 1024 and isinstance(v[0], float)
    # a2: empty -> np.array([]) with shape (0,) (246-247); order kept; blank rows are zero
    e = run(embedding.embed_texts_in_batches([]))
    assert isinstance(e, np.ndarray) and e.shape == (0,)
    texts = ["alpha beta", "", "gamma", "  ", "delta epsilon zeta"]
    e = run(embedding.embed_texts_in_batches(texts, batch_size=2))
    assert e.shape == (5, 1024) and e.dtype == np.float32 and e.flags["C_CONTIGUOUS"]
    assert np.all(e[1] == 0) and np.all(e[3] == 0) and np.any(e[0] != 0)
    assert np.array_equal(e[2], np.asarray(run(embedding.ollama_embed_text("gamma")), dtype=np.float32))
    # blanks never reach the encoder; the three non-blank texts of one call go out as ONE encoder
    # batch whatever batch_size says (the encoder batches for the GPU on its own)
    assert [len(c) for c in shim.calls[:3]] == [1, 3, 1]
    # a3: blank query -> size 0; else [1, dim] fp32, no prompt prefix
    assert run(embedding.embed_query(" ")).size == 0
    q = run(embedding.embed_query("gamma"))
    assert q.shape == (1, 1024) and q.dtype == np.float32 and np.array_equal(q[0], e[2])


def test_no_encoder_raises_like_the_reference(monkeypatch):
    embedding.set_embedder(None)
    monkeypatch.setattr(config, "RASS_MODEL_DIR", "")
    with pytest.raises(RuntimeError):
        run(embedding.ollama_embed_text("x"))  # main.py:235 raises too (raise_for_status)


def test_store_then_semantic_search_roundtrip(shim):
    name = "idx-user1"
    docs = _docs(40, patient="p1") + _docs(40, patient="p2", prefix="md-note")
    structured = [{"doc_id": "Patient-1", "doc_type": "structured", "patientId": "p1", "patientName": "A B"}]
    run(indexer.store_fhir_docs_in_opensearch(structured, docs, None, name))
    ix = indexer.HipIndexer(None, name)
    assert ix.has_any_data()
    assert not indexer.HipIndexer(None, "idx-nobody").has_any_data()

    q = run(embedding.embed_query("chunk number 3 about topic3"))
    hits = ix.semantic_search(query_emb=q, k=5, filter_clause=None, patient_id=None, query="ignored")  # ask() kwargs
    assert len(hits) == 5 and all(isinstance(h, tuple) and isinstance(h[1], float) for h in hits)
    assert hits[0][0]["unstructuredText"] == "chunk number 3 about topic3"
    assert hits[0][1] >= hits[1][1] >= hits[-1][1]
    assert abs(hits[0][1] - 1.0) < 1e-5          # opensearch score of cos=1 is 1/(2-1)
    assert "embedding" not in hits[0][0]          # quirk 5 not replicated by default

    # term filter on patientId (1549) is an exact pre-filter
    hits = ix.semantic_search(q, k=10, patient_id="p2")
    assert len(hits) == 10 and all(h[0]["patientId"] == "p2" for h in hits)
    hits = ix.semantic_search(q, k=3, filter_clause={"term": {"patientId": "p1"}})
    assert all(h[0]["patientId"] == "p1" for h in hits)
    assert ix.semantic_search(q, k=3, patient_id="never-indexed") == []
    # ask() passes the NER entity LIST as filter_clause (quirk 1): tolerated, not a filter
    assert len(ix.semantic_search(q, k=3, filter_clause=[{"text": "x", "label": "Y"}])) == 3
    # empty query embedding -> [] (1534-1535)
    assert ix.semantic_search(np.array([]), k=3) == []
    # structured docs carry no embedding and are never returned by k-NN
    all_hits = ix.semantic_search(q, k=32)
    assert all(h[0]["doc_type"] == "unstructured" for h in all_hits)


def test_scores_match_bruteforce_and_modes(shim, monkeypatch):
    from oracle import oracle as O
    name = "idx-user2"
    docs = _docs(100)
    run(indexer.store_fhir_docs_in_opensearch([], docs, None, name))
    ix = indexer.HipIndexer(None, name)
    q = run(embedding.embed_query("about topic5 chunk"))
    emb = run(embedding.embed_texts_in_batches([d["unstructuredText"] for d in docs]))
    xn = O.normalize_ref(emb).astype(np.float32)
    rs, ri = O.search(xn, O.normalize_ref(q).astype(np.float32), 7)
    monkeypatch.setattr(config, "RASS_SCORE_MODE", "cosine")
    hits = ix.semantic_search(q, k=7)
    assert [h[0]["doc_id"] for h in hits] == [docs[i]["doc_id"] for i in ri[0]]
    assert np.allclose([h[1] for h in hits], rs[0], atol=1e-6)
    monkeypatch.setattr(config, "RASS_SCORE_MODE", "opensearch")
    hits_os = ix.semantic_search(q, k=7)
    assert np.allclose([h[1] for h in hits_os], 1.0 / (2.0 - rs[0]), atol=1e-6)
    # hybrid's knn clause carries boost 2.0 (1595); blank query text -> [] (1570)
    hy = ix.hybrid_search("some text", q, k=7)
    assert np.allclose([h[1] for h in hy], 2.0 / (2.0 - rs[0]), atol=1e-6)
    assert ix.hybrid_search("  ", q, k=7) == []
    assert np.allclose([h[1] for h in ix.knn_scores(q, k=7, boost=1.5)], 1.5 / (2.0 - rs[0]), atol=1e-6)


def test_doc_id_overwrite_semantics(shim):
    name = "idx-user3"
    docs = _docs(10)
    run(indexer.store_fhir_docs_in_opensearch([], docs, None, name))
    st = REGISTRY.get(name)
    changed = dict(docs[4], unstructuredText="completely different words now")
    run(indexer.store_fhir_docs_in_opensearch([], [changed], None, name))
    assert st.index.rows == 11 and st.index.count == 10 and len(st.doc_row) == 10
    ix = indexer.HipIndexer(None, name)
    q_old = run(embedding.embed_query(docs[4]["unstructuredText"]))
    ids = [h[0]["doc_id"] for h in ix.semantic_search(q_old, k=10)]
    assert ids.count("text-note-4") == 1
    q_new = run(embedding.embed_query("completely different words now"))
    top = ix.semantic_search(q_new, k=1)[0][0]
    assert top["doc_id"] == "text-note-4" and top["unstructuredText"] == "completely different words now"
    # duplicate doc_id inside one batch: the last one wins
    run(indexer.store_fhir_docs_in_opensearch([], [dict(docs[1], unstructuredText="first version"),
                                                   dict(docs[1], unstructuredText="second version")], None, name))
    assert st.index.count == 10
    assert st.row_doc[st.doc_row["text-note-1"]]["unstructuredText"] == "second version"


def test_search_errors_are_swallowed_like_the_reference(shim, caplog):
    name = "idx-user4"
    run(indexer.store_fhir_docs_in_opensearch([], _docs(5), None, name))
    st = REGISTRY.get(name)

    def boom(*a, **k):
        raise RuntimeError("device lost")
    st.index.search = boom
    ix = indexer.HipIndexer(None, name)
    q = run(embedding.embed_query("chunk"))
    assert ix.semantic_search(q, k=3) == []          # 1558-1560: log + []
    assert any("Semantic search error" in r.message for r in caplog.records)


def test_install_rebinds_reference_names(shim):
    import types
    fake_main = types.SimpleNamespace(OpenSearchIndexer=object, ensure_index_exists=None, embed_query=None,
                                      embed_texts_in_batches=None, ollama_embed_text=None,
                                      store_fhir_docs_in_opensearch=None, chunk_text=lambda t: [t])
    keep = fake_main.chunk_text
    indexer.install(fake_main)
    assert fake_main.OpenSearchIndexer is indexer.HipIndexer
    assert fake_main.embed_query is embedding.embed_query
    assert fake_main.store_fhir_docs_in_opensearch is indexer.store_fhir_docs_in_opensearch
    assert fake_main.chunk_text is keep  # unchanged by contract


def test_patient_dictionary():
    from rassengine_amd.docstore import PatientDictionary
    d = PatientDictionary()
    assert d.encode(None) == 0 and d.encode("") == 0
    a, b = d.encode("p-1"), d.encode(77)
    assert (a, b) == (1, 2) and d.encode("p-1") == 1 and d.lookup("77") == 2 and d.lookup("zz") is None and len(d) == 2
