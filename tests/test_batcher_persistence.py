"""CPU tests: the cross-request query micro-batcher and the metadata persistence, on the
oracle-backed index double (the HIP index has the same surface)."""
import asyncio

import numpy as np
import pytest

from rassengine_amd import embedding, indexer
from rassengine_amd.batcher import QueryBatcher
from rassengine_amd.docstore import REGISTRY, IndexState
from tests.helpers import HashEmbedder, OracleIndex


class CountingIndex(OracleIndex):
    def __init__(self, dim):
        super().__init__(dim)
        self.calls = []

    def search(self, queries, k, q_filter=None, q_filter_mask=None):
        self.calls.append((queries.shape[0], k, None if q_filter is None else q_filter.copy()))
        return super().search(queries, k, q_filter, q_filter_mask)


def test_batcher_coalesces_and_matches_individual_searches():
    rng = np.random.default_rng(0)
    idx = CountingIndex(64)
    idx.add(rng.standard_normal((500, 64), dtype=np.float32), tags=rng.integers(1, 4, size=500).astype(np.int32))
    qs = rng.standard_normal((100, 64), dtype=np.float32)
    ks = rng.integers(1, 11, size=100)
    codes = rng.integers(-1, 4, size=100)

    async def main():
        b = QueryBatcher(idx, max_batch=32, max_delay_ms=20.0)
        res = await asyncio.gather(*[b.search(qs[i], int(ks[i]), int(codes[i])) for i in range(100)])
        await b.close()
        return b, res

    b, res = asyncio.run(main())
    assert b.served == 100 and b.scans <= 8          # >= 13x fewer scans than requests
    assert all(c[0] <= 32 for c in idx.calls)
    ref = OracleIndex(64)
    ref._rows, ref._tags = idx._rows, idx._tags
    for i, (s, ids) in enumerate(res):
        f = None if codes[i] < 0 else np.array([codes[i]], dtype=np.int32)
        rs, ri = ref.search(qs[i:i + 1], int(ks[i]), f)
        assert ids.shape == (ks[i],) and np.array_equal(ids, ri[0]) and np.allclose(s, rs[0], atol=1e-6)


def test_batcher_single_request_latency_bound_and_errors():
    idx = CountingIndex(16)
    idx.add(np.eye(16, dtype=np.float32))

    async def one():
        b = QueryBatcher(idx, max_delay_ms=1.0)
        s, i = await b.search(np.eye(16, dtype=np.float32)[3], 2)
        await b.close()
        return s, i

    s, i = asyncio.run(one())
    assert i[0] == 3 and idx.calls[-1][0] == 1

    class Boom:
        def search(self, *a, **k):
            raise RuntimeError("device lost")

    async def failing():
        b = QueryBatcher(Boom(), max_delay_ms=1.0)
        with pytest.raises(RuntimeError):
            await b.search(np.zeros(4, dtype=np.float32), 1)
        await b.close()

    asyncio.run(failing())
    with pytest.raises(ValueError):
        QueryBatcher(idx, max_batch=64)


def test_async_semantic_search_shares_scans():
    REGISTRY.clear()
    idx = CountingIndex(1024)
    REGISTRY.set_index_factory(lambda name: idx)
    emb = HashEmbedder(1024)
    embedding.set_embedder(emb)
    try:
        docs = [{"doc_id": f"d{i}", "doc_type": "unstructured", "patientId": f"p{i % 2}",
                 "unstructuredText": f"note {i} topic{i % 9}"} for i in range(120)]

        async def main():
            await indexer.store_fhir_docs_in_opensearch([], docs, None, "idx-b")
            ix = indexer.HipIndexer(None, "idx-b")
            qs = [await embedding.embed_query(f"note {i} topic{i % 9}") for i in range(40)]
            idx.calls.clear()
            out = await asyncio.gather(*[ix.asemantic_search(query_emb=qs[i], k=3, patient_id=f"p{i % 2}", query="x")
                                         for i in range(40)])
            sync = [ix.semantic_search(qs[i], k=3, patient_id=f"p{i % 2}") for i in range(40)]
            await REGISTRY.get("idx-b").batcher.close()
            return out, sync

        out, sync = asyncio.run(main())
        assert len(idx.calls) - 40 <= 14                     # 40 async requests share scans (+40 sync calls)
        for a, b in zip(out, sync):
            assert [d["doc_id"] for d, _ in a] == [d["doc_id"] for d, _ in b]
            assert np.allclose([s for _, s in a], [s for _, s in b])
        assert out[5][0][0]["doc_id"] == "d5"
    finally:
        embedding.set_embedder(None)
        REGISTRY.set_index_factory(None)
        REGISTRY.clear()


def test_metadata_persistence_roundtrip(tmp_path):
    class SavingIndex(OracleIndex):
        def save(self, path):
            with open(path, "wb") as f:
                np.savez(f, rows=self._rows, tags=self._tags)

    def loader(name, path):
        z = np.load(path)
        i = SavingIndex(z["rows"].shape[1])
        i._rows, i._tags = z["rows"], z["tags"]
        return i

    REGISTRY.clear()
    REGISTRY.set_index_factory(lambda name: SavingIndex(1024))
    embedding.set_embedder(HashEmbedder(1024))
    try:
        docs = [{"doc_id": f"d{i}", "doc_type": "unstructured", "patientId": f"p{i % 3}",
                 "unstructuredText": f"text {i} alpha{i % 5}"} for i in range(30)]
        structured = [{"doc_id": "Patient-1", "doc_type": "structured", "patientId": "p1"}]
        asyncio.run(indexer.store_fhir_docs_in_opensearch(structured, docs, None, "idx-p"))
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], [dict(docs[4], unstructuredText="rewritten")], None, "idx-p"))
        st = REGISTRY.get("idx-p")
        prefix = str(tmp_path / "shard0")
        st.save(prefix)
        st2 = IndexState.load("idx-p", prefix, loader)
        assert st2.doc_row == st.doc_row and st2.structured == st.structured
        assert st2.patients.lookup("p2") == st.patients.lookup("p2") and len(st2.patients) == 3
        assert st2.doc_types.lookup("unstructured") == st.doc_types.lookup("unstructured") == 1

        # crash safety: a second save writes a NEW generation and only then drops the old one;
        # a save that dies half-way leaves the previous manifest + vector file intact and loadable
        import json
        import os
        files1 = sorted(os.listdir(tmp_path))
        assert files1 == ["shard0.g000001.rass", "shard0.meta.json"]
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], [dict(docs[5], unstructuredText="again")], None, "idx-p"))

        class Dying(SavingIndex):
            def save(self, path):
                with open(path, "wb") as f:
                    f.write(b"partial")
                raise OSError("disk full")
        good = st.index
        dying = Dying(1024)
        dying._rows, dying._tags = good._rows, good._tags
        st.index = dying
        with pytest.raises(OSError):
            st.save(prefix)
        st.index = good
        assert json.load(open(prefix + ".meta.json"))["vectors"] == "shard0.g000001.rass"
        assert IndexState.load("idx-p", prefix, loader).doc_row == st2.doc_row      # old generation still loads
        st.save(prefix)
        assert sorted(f for f in os.listdir(tmp_path) if not f.endswith(".tmp")) == ["shard0.g000002.rass",
                                                                                      "shard0.meta.json"]
        st3 = IndexState.load("idx-p", prefix, loader)
        assert st3.generation == 2 and st3.row_doc[st3.doc_row["d5"]]["unstructuredText"] == "again"

        # a state built WITHOUT load() (generation 0) saving over a live manifest must not reuse the file
        # name that manifest references (it would be overwritten before the new manifest is committed)
        fresh = IndexState("idx-p", good)
        fresh.row_doc, fresh.doc_row, fresh.structured = list(st.row_doc), dict(st.doc_row), dict(st.structured)
        for p in st.patients.names():
            fresh.patients.encode(p)
        for t in st.doc_types.names():
            fresh.doc_types.encode(t)
        assert fresh.generation == 0
        fresh.save(prefix)
        assert fresh.generation == 3 and json.load(open(prefix + ".meta.json"))["vectors"] == "shard0.g000003.rass"
        assert sorted(os.listdir(tmp_path)) == ["shard0.g000003.rass", "shard0.meta.json"]
        st.generation = 3

        # a manifest that does not belong to its vector file is rejected, not silently loaded
        meta = json.load(open(prefix + ".meta.json"))
        meta["row_doc"] = meta["row_doc"][:-1]
        json.dump(meta, open(prefix + ".meta.json", "w"))
        with pytest.raises(ValueError, match="disagree"):
            IndexState.load("idx-p", prefix, loader)
        meta["row_doc"] = st.row_doc
        json.dump(meta, open(prefix + ".meta.json", "w"))
        REGISTRY.clear()
        REGISTRY.put(st2)
        ix = indexer.HipIndexer(None, "idx-p")
        q = asyncio.run(embedding.embed_query("rewritten"))
        hits = ix.semantic_search(q, k=2, patient_id="p1")
        assert hits[0][0]["doc_id"] == "d4" and hits[0][0]["unstructuredText"] == "rewritten"
        assert ix.has_any_data()
    finally:
        embedding.set_embedder(None)
        REGISTRY.set_index_factory(None)
        REGISTRY.clear()


def test_incremental_persistence_replays_deltas(tmp_path):
    """IndexState.save_delta: O(delta) segments on top of a snapshot — appended rows (the stored bits), their docs, overwrites
    of snapshot rows (tombstones), overwrites inside a delta, structured docs; load() = snapshot + segments in order; a stray
    segment a crash left unlisted is ignored; the next snapshot clears the log."""
    import json
    import os

    class SavingIndex(OracleIndex):
        def save(self, path):
            with open(path, "wb") as f:
                np.savez(f, rows=self._rows, tags=self._tags)

    def loader(name, path):
        z = np.load(path)
        i = SavingIndex(z["rows"].shape[1])
        i._rows, i._tags = z["rows"], z["tags"]
        return i

    def docs_of(lo, hi, word="text"):
        return [{"doc_id": f"d{i}", "doc_type": "unstructured", "patientId": f"p{i % 4}",
                 "unstructuredText": f"{word} {i} alpha{i % 5}"} for i in range(lo, hi)]

    def same_state(a, b, query="text 33 alpha3"):
        assert a.doc_row == b.doc_row and a.structured == b.structured and a.row_doc == b.row_doc
        assert a.index.rows == b.index.rows and a.index.count == b.index.count
        assert np.array_equal(a.index._rows, b.index._rows) and np.array_equal(a.index._tags, b.index._tags)
        assert a.patients.names() == b.patients.names() and a.doc_types.names() == b.doc_types.names()
        q = HashEmbedder(1024).encode([query])
        ra, rb = a.index.search(q, 7), b.index.search(q, 7)
        assert np.array_equal(ra[1], rb[1]) and np.array_equal(ra[0], rb[0])

    REGISTRY.clear()
    REGISTRY.set_index_factory(lambda name: SavingIndex(1024))
    embedding.set_embedder(HashEmbedder(1024))
    try:
        name, prefix = "idx-inc", str(tmp_path / "inc")
        store = lambda s, u: asyncio.run(indexer.store_fhir_docs_in_opensearch(s, u, None, name))
        store([{"doc_id": "Patient-1", "doc_type": "structured", "patientId": "p1"}], docs_of(0, 30))
        st = REGISTRY.get(name)
        assert st.save_delta(prefix) is False                   # no snapshot yet: the caller must save()
        st.save(prefix)
        assert st.save_delta(prefix) is True                    # nothing changed: nothing written
        assert not os.path.exists(prefix + ".deltas.json")
        # delta 1: new rows (a new patient code among them), an overwrite of a snapshot row, an overwrite inside the delta
        store([{"doc_id": "Patient-2", "doc_type": "structured", "patientId": "p9"}],
              docs_of(30, 40) + [dict(docs_of(4, 5)[0], unstructuredText="rewritten four")] +
              [{"doc_id": "d99", "doc_type": "note", "patientId": "p9", "unstructuredText": "first"},
               {"doc_id": "d99", "doc_type": "note", "patientId": "p9", "unstructuredText": "second wins"}])
        assert st.save_delta(prefix) is True
        seg1 = json.load(open(prefix + ".deltas.json"))["segments"]
        assert len(seg1) == 1 and os.path.getsize(os.path.join(str(tmp_path), seg1[0])) < 14 * 4096 + 8192   # 13 rows, not 43
        same_state(st, IndexState.load(name, prefix, loader))
        # delta 2: overwrite a row that lives in delta 1, rewrite a structured doc
        store([{"doc_id": "Patient-1", "doc_type": "structured", "patientId": "p1", "gender": "f"}],
              [dict(docs_of(33, 34)[0], unstructuredText="thirty-three again")])
        assert st.save_delta(prefix) is True
        st2 = IndexState.load(name, prefix, loader)
        same_state(st, st2)
        assert st2.structured["Patient-1"]["gender"] == "f" and st2.doc_row["d99"] == st.doc_row["d99"]
        assert len(json.load(open(prefix + ".deltas.json"))["segments"]) == 2
        # the loaded state keeps appending to the same log
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs_of(50, 53), None, name))      # on the live state
        REGISTRY.put(st2)
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs_of(50, 53), None, name))      # and on the restored one
        assert st2.save_delta(prefix) is True
        same_state(st2, IndexState.load(name, prefix, loader))
        # a segment a crash left behind without listing it is ignored; a listing of another generation too
        open(os.path.join(str(tmp_path), "inc.g000001.d000009.delta"), "wb").write(b"garbage")
        same_state(st2, IndexState.load(name, prefix, loader))
        # a torn / reordered log is an error, not silent corruption
        listing = json.load(open(prefix + ".deltas.json"))
        json.dump({"generation": listing["generation"], "segments": listing["segments"][1:]}, open(prefix + ".deltas.json", "w"))
        with pytest.raises(ValueError):
            IndexState.load(name, prefix, loader)
        json.dump(listing, open(prefix + ".deltas.json", "w"))
        # the next snapshot folds the log in and clears it
        st2.save(prefix)
        assert not os.path.exists(prefix + ".deltas.json") and not [f for f in os.listdir(str(tmp_path)) if f.endswith(".delta")]
        same_state(st2, IndexState.load(name, prefix, loader))
    finally:
        embedding.set_embedder(None)
        REGISTRY.set_index_factory(None)
        REGISTRY.clear()
