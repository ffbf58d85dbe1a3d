"""The drop-in boundary pinned by what the REFERENCE'S OWN CODE emits: tests/golden/reference_boundary.json
holds the request bodies / return values of ``OpenSearchIndexer``'s four knn-bearing builders (app/main.py:1527-1560,
1562-1615, 1710-1775, 1962-2027) and the bulk actions of ``store_fhir_docs_in_opensearch`` (1211-1282), produced by executing
those definitions (AST-lifted, tests/golden/make_reference_boundary_fixtures.py) against recording fakes.

Every arithmetic / bookkeeping step the reference DOES hold on this path is checked against it: the normalised query vector
the scan sees, ``size`` / ``k``, which term filters apply (AND of the list), ``routing`` = the patient filter, the knn clause's
boost, ``[]`` on an empty embedding / blank text / a failing backend; on the write side ``_id`` / ``_routing`` / overwrite
order / the slicing into bulks of BATCH_SIZE and the normalised ``embedding`` rows vs what the index stores.  (The k-NN
SCORES live in OpenSearch / nmslib and stay unpinned: SURVEY §8c.)  CPU here (oracle-backed index double); the ``gpu``-marked
tests repeat the arithmetic on the HIP engine."""
import asyncio
import json
import os

import numpy as np
import pytest

from rassengine_amd import config, embedding, indexer
from rassengine_amd.docstore import REGISTRY, TAG_DOCTYPE_MASK, TAG_DOCTYPE_SHIFT, TAG_PATIENT_MASK
from tests import fake_reference
from tests.helpers import OracleIndex

HERE = os.path.dirname(os.path.abspath(__file__))
FX = json.load(open(os.path.join(HERE, "golden", "reference_boundary.json"), encoding="utf-8"))
BOOST = {"semantic_search": 1.0, "hybrid_search": indexer.BOOST_HYBRID, "hybrid_structured_search": indexer.BOOST_HYBRID_STRUCTURED,
         "multi_intent_search": indexer.BOOST_MULTI_INTENT}


def _q(name):
    return np.asarray(FX["queries"][name], dtype=np.float32)


def _ulp_diff(a, b):
    """Distance in units in the last place between two float32 arrays (same sign assumed where it matters)."""
    ai = np.asarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    bi = np.asarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


class RecordingIndex(OracleIndex):
    def __init__(self, dim):
        super().__init__(dim)
        self.calls = []
        self.fail = False

    def search(self, queries, k, q_filter=None, q_filter_mask=None):
        self.calls.append({"q": np.array(queries, copy=True), "k": int(k),
                           "filter": None if q_filter is None else int(q_filter[0]),
                           "mask": None if q_filter_mask is None else int(q_filter_mask[0])})
        if self.fail:
            raise RuntimeError("connection refused")
        return super().search(queries, k, q_filter, q_filter_mask)


def _term_filters(req, method):
    """The term filters of a recorded request as {field: [values]} (non-dict entries — the NER list ask() passes,
    SURVEY §8b quirk 1 — are not term filters; the engine ignores them, documented)."""
    out = {}
    for f in req["filter"] or []:
        if isinstance(f, dict) and isinstance(f.get("term"), dict):
            for field, val in f["term"].items():
                out.setdefault(field, []).append(val)
    return out


@pytest.fixture
def served():
    REGISTRY.clear()
    fake_reference.reset()
    mode0, config.RASS_KNN_PREFETCH = config.RASS_KNN_PREFETCH, 0
    score0, config.RASS_SCORE_MODE = config.RASS_SCORE_MODE, "cosine"
    made = {}

    def factory(name):
        made[name] = RecordingIndex(made.get("__dim__", FX["dim"]))
        return made[name]
    REGISTRY.set_index_factory(factory)
    yield made
    config.RASS_KNN_PREFETCH, config.RASS_SCORE_MODE = mode0, score0
    embedding.set_embedder(None)
    REGISTRY.set_index_factory(None)
    REGISTRY.clear()


def _populate(name, dim, rng):
    docs = [{"doc_id": f"d-{i}", "doc_type": "unstructured" if i % 4 else "structured", "patientId": f"p{i % 3}",
             "unstructuredText": f"t{i}"} for i in range(60)]
    indexer.add_documents(name, docs, rng.standard_normal((60, dim)).astype(np.float32))
    return docs


def test_fixture_is_what_the_generator_describes():
    assert FX["top_k_default"] == config.TOP_K == 3 and FX["batch_size"] == config.BATCH_SIZE == 64
    assert len(FX["search"]) == 55 and len(FX["store"]["actions"]) == 72
    methods = {c["method"] for c in FX["search"]}
    assert methods == set(BOOST)


def test_reference_emitted_vector_is_the_oracles_normalise(oracle):
    """a4 (app/main.py:1536-1537): every ``vector`` the reference sent == normalize_ref(query)[0], BIT FOR BIT."""
    n = 0
    for c in FX["search"]:
        for req in c.get("requests") or []:
            want = np.asarray(req["knn"]["vector"], dtype=np.float32)
            got = oracle.normalize_ref(_q(c["query"])).astype(np.float32)[0]          # row 0 only: (...)[0].tolist()
            assert want.shape == got.shape and np.array_equal(want.view(np.uint32), got.view(np.uint32)), (c["method"], c["query"])
            n += 1
    assert n >= 40


def test_knn_builders_issue_the_reference_request(served, oracle):
    rng = np.random.default_rng(0)
    for dim, qnames in ((FX["dim"], ("q_scaled", "q_tiny", "q_two_rows")), (1024, ("q_1024",))):
        served["__dim__"] = dim
        name = f"rass-idx-user1-{dim}"
        _populate(name, dim, rng)
        st = REGISTRY.get(name)
        idx = served[name]
        ix = indexer.HipIndexer(None, name)
        for c in FX["search"]:
            if c.get("query") not in qnames:
                continue
            method, kw = c["method"], dict(c["kwargs"])
            args = ((c["text"],) if c["text"] is not None else ()) + (_q(c["query"]),)
            idx.calls.clear()
            hits = getattr(ix, method)(*args, **kw)
            if c["raised"]:
                # quirk 3 (app/main.py:1764): the reference's hybrid_structured_search raises KeyError without a filter /
                # patient; tolerated, not replicated — the doc_type filter is applied on its own
                assert c["raised"] == "KeyError" and method == "hybrid_structured_search" and isinstance(hits, list)
                continue
            req = c["requests"][0]
            terms = _term_filters(req, method)
            contradictory = any(len(set(map(str, v))) > 1 for v in terms.values())
            if contradictory:               # the reference ANDs e.g. doc_type = unstructured AND doc_type = structured
                assert hits == [] and idx.calls == []
                continue
            assert len(idx.calls) == 1, (method, kw)
            call = idx.calls[0]
            # size == k == what the scan is asked for; the scan sees row 0 of the caller's embedding, un-normalised (the
            # engine normalises: the double with normalize_ref, the HIP index with its kernel — pinned above / on the GPU)
            assert call["k"] == req["size"] == req["knn"]["k"] == req["terminate_after"]
            assert call["q"].shape == (1, dim) and np.array_equal(call["q"][0], _q(c["query"])[0])
            # filters: the AND of the reference's term filters == the masked tag compare the scan runs
            want_pid = terms.get("patientId", [None])[0]
            want_dt = terms.get("doc_type", [None])[0]
            if want_pid is None and want_dt is None:
                assert call["filter"] is None
            else:
                got_pid = got_dt = None
                if call["mask"] & TAG_PATIENT_MASK:
                    got_pid = st.patients.names()[(call["filter"] & TAG_PATIENT_MASK) - 1]
                if call["mask"] & TAG_DOCTYPE_MASK:
                    got_dt = st.doc_types.names()[((call["filter"] & TAG_DOCTYPE_MASK) >> TAG_DOCTYPE_SHIFT) - 1]
                assert (got_pid, got_dt) == (want_pid, want_dt), (method, kw)
            # routing = patient_id (1552): the rows of that patient are the only ones searched
            if req["routing"] is not None:
                assert req["routing"] == kw.get("patient_id") == want_pid
                assert hits and all(d["patientId"] == req["routing"] for d, _ in hits)
            # the knn clause's boost multiplies the hit's score (raw cosine mode: score = boost x cos)
            assert (req["knn"]["boost"] or 1.0) == BOOST[method]
            xn, qn = idx._rows, oracle.normalize_ref(_q(c["query"])[:1]).astype(np.float32)
            for d, s in hits:
                row = st.doc_row[d["doc_id"]]
                assert abs(s - BOOST[method] * float(xn[row] @ qn[0])) <= 1e-5
            assert len(hits) <= call["k"] and [s for _, s in hits] == sorted([s for _, s in hits], reverse=True)


def test_empty_blank_and_failing_backend_answer_like_the_reference(served):
    rng = np.random.default_rng(1)
    _populate("rass-idx-user1", FX["dim"], rng)
    idx = served["rass-idx-user1"]
    ix = indexer.HipIndexer(None, "rass-idx-user1")
    seen = set()
    for c in FX["search"]:
        case = c.get("case")
        if case is None:
            continue
        seen.add(case)
        q = np.array([]) if case == "empty_embedding" else _q("q_scaled")
        args = ((c["text"],) if c["text"] is not None else ()) + (q,)
        idx.fail = case == "client_raises"
        idx.calls.clear()
        got = getattr(ix, c["method"])(*args)
        idx.fail = False
        if c["raised"]:      # quirk 3 again (KeyError before the request is even sent)
            assert c["method"] == "hybrid_structured_search" and got == []
            continue
        assert got == c["returned"] == [], (c["method"], case)
        if case != "client_raises":
            assert c["n_requests"] == 0 and idx.calls == []          # no request leaves for an empty embedding / blank text
        else:
            assert c["n_requests"] == 1 and len(idx.calls) == 1
    assert seen == {"empty_embedding", "blank_text", "client_raises"}
    # has_any_data (1470-1478): count > 0; False on no client-side data
    assert [h["out"] for h in FX["has_any_data"]] == [True, False, False, False]
    assert ix.has_any_data() is True and indexer.HipIndexer(None, "never-created").has_any_data() is False


def _store(module, client, name):
    S = FX["store"]
    raw = np.asarray(S["raw_embeddings"], dtype=np.float32)
    calls = []

    async def fake_embed(texts, batch_size=config.BATCH_SIZE):
        calls.append((len(texts), batch_size))
        return raw[:len(texts)].copy()
    module.embed_texts_in_batches = fake_embed
    docs = json.loads(json.dumps(S["unstructured_docs"]))
    asyncio.run(indexer.store_fhir_docs_in_opensearch(json.loads(json.dumps(S["structured_docs"])), docs, client, name,
                                                      embed_fn=fake_embed))
    return calls


def _check_store(st, index, max_ulp):
    S = FX["store"]
    acts = S["actions"]
    assert S["bulk_sizes"] == [2, 64, 6] and S["no_client_bulks"] == 0
    assert all(a["_op_type"] == "index" and a["_index"] == "rass-idx-user1" for a in acts)
    un = [a for a in acts if a["doc_type"] == "unstructured"]
    assert len(un) == 70 and [a["_id"] for a in acts[:2]] == ["Condition-1", "Observation-9"]
    assert sorted(st.structured) == sorted(a["_id"] for a in acts[:2])
    last = {}
    for a in un:
        last[a["_id"]] = a                                  # _id = doc_id: a later action overwrites (1253-1265)
    assert len(last) == 69 and set(st.doc_row) == set(last) and index.count == 69 and index.rows == 70
    worst = 0
    for doc_id, a in last.items():
        row = st.doc_row[doc_id]
        got = np.asarray(index.get_row(row), dtype=np.float32)
        want = np.asarray(a["embedding"], dtype=np.float32)   # the reference's e / (||e|| + 1e-9), as it bulk-indexed it
        assert got.shape == want.shape
        if not want.any():
            assert not got.any()                            # the blank chunk: a zero row stays a zero row
        else:
            worst = max(worst, int(_ulp_diff(got, want).max()))
        assert st.row_doc[row]["patientId"] == a["_routing"]                 # _routing = patientId (1263)
        code = st.patients.lookup(a["_routing"])
        assert (code or 0) == (st.tag_of(st.row_doc[row]) & TAG_PATIENT_MASK)
    assert worst <= max_ulp, worst
    # the overwritten action's row is a tombstone: its vector never comes back
    dup_first = [a for a in un if a["_id"] == "text-note-5"][0]
    q = np.asarray(dup_first["embedding"], dtype=np.float32)[None, :]
    s, i = index.search(q, 1)
    assert int(i[0, 0]) != 5 and st.row_doc[5] is None
    return worst


def test_write_side_stores_what_the_reference_bulk_indexes(served):
    indexer._ORIGINALS.clear()
    m = fake_reference.make_module()
    indexer.install(m)
    try:
        calls = _store(m, None, "rass-idx-user1")
        assert calls == [(e["n_texts"], e["batch_size"]) for e in FX["store"]["embed_calls"]] == [(70, 64)]
        assert fake_reference.BULKED == []                 # no client: nothing is bulk-indexed (the reference returns early)
        _check_store(REGISTRY.get("rass-idx-user1"), served["rass-idx-user1"], max_ulp=0)
    finally:
        indexer.uninstall(m)


def test_kept_text_engine_receives_the_reference_actions_without_vectors(served):
    """With a text engine kept (client given), the bulk actions the shim forwards are the reference's — same ``_op_type`` /
    ``_index`` / ``_id`` / ``_routing``, same order, same slicing into bulks of BATCH_SIZE — minus the 1024-float
    ``embedding`` (the vectors live in HBM)."""
    indexer._ORIGINALS.clear()              # records other tests' modules left behind
    m = fake_reference.make_module()
    indexer.install(m)
    sizes = []
    rec = indexer._ORIGINALS[id(m)]
    orig_bulk = rec["bulk"]

    def counting_bulk(client, actions):
        sizes.append(len(actions))
        return orig_bulk(client, actions)
    rec["bulk"] = counting_bulk
    try:
        _store(m, fake_reference.FakeClient(), "rass-idx-user1")
        want = FX["store"]["actions"]
        got = fake_reference.BULKED
        assert sizes == FX["store"]["bulk_sizes"]
        assert [(a["_op_type"], a["_index"], a["_id"], a["_routing"]) for a in got] == \
               [(a["_op_type"], a["_index"], a["_id"], a["_routing"]) for a in want]
        assert all("embedding" not in a["_source"] for a in got)
    finally:
        indexer.uninstall(m)


# ------------------------------------------------------------------------------------------------ the HIP engine
@pytest.mark.gpu
def test_hip_normalise_matches_the_reference_vector(gpu):
    """The vector the HIP scan sees (rass::normalize_rows of the caller's embedding) vs the ``vector`` the reference sent:
    within 2 ulp (the kernel's reduction order differs from numpy's)."""
    import torch
    from rassengine_amd import ops
    worst, n = 0, 0
    for c in FX["search"]:
        for req in c.get("requests") or []:
            want = np.asarray(req["knn"]["vector"], dtype=np.float32)
            q = torch.from_numpy(_q(c["query"])[:1]).cuda()
            got = ops.normalize_rows(q).cpu().numpy()[0][:want.size]
            worst = max(worst, int(_ulp_diff(got, want).max()))
            n += 1
    assert n >= 40 and worst <= 2, worst


@pytest.mark.gpu
def test_hip_index_stores_the_reference_embedding_rows(gpu):
    """store_fhir_docs_in_opensearch over the REAL index: ``rass_index_get_row`` of every live doc vs the normalised
    ``embedding`` the reference bulk-indexed for that ``_id`` (<= 2 ulp), overwrite order, ``_routing`` = the row's tag."""
    from rassengine_amd.engine import Engine
    REGISTRY.clear()
    fake_reference.reset()
    eng = Engine(0, FX["dim"])
    REGISTRY.set_index_factory(lambda name: eng.open_index(name))
    mode0, config.RASS_KNN_PREFETCH = config.RASS_KNN_PREFETCH, 0
    m = fake_reference.make_module()
    indexer.install(m)
    try:
        _store(m, None, "rass-idx-user1")
        st = REGISTRY.get("rass-idx-user1")
        worst = _check_store(st, st.index, max_ulp=2)
        print(f"stored rows vs the reference's bulk-indexed embeddings: worst {worst} ulp")
        # and the read side on the same index: routing / filters / k of three recorded requests
        ix = indexer.HipIndexer(None, "rass-idx-user1")
        for c in FX["search"]:
            if c.get("query") != "q_scaled" or c["method"] != "semantic_search" or c["raised"]:
                continue
            hits = ix.semantic_search(_q("q_scaled"), **c["kwargs"])
            req = c["requests"][0]
            assert len(hits) <= req["size"]
            if req["routing"] is not None:
                assert hits and all(d["patientId"] == req["routing"] for d, _ in hits)
    finally:
        indexer.uninstall(m)
        config.RASS_KNN_PREFETCH = mode0
        REGISTRY.set_index_factory(None)
        REGISTRY.clear()
        eng.close()
