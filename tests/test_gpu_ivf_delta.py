"""IVF behind the boundary (VERDICT r3 #2b): an ``ivf.IvfBackedIndex`` keeps taking rows after its IVF was built — they
form a flat DELTA that every search scans exactly next to the probe (``rass_ivf_search_delta``) — and honours
overwrites / deletes on both sides.  The reference's index is approximate (HNSW, app/main.py:563-572) and
incrementally insertable (bulk per 64 docs, app/main.py:1253-1282).

Pinned: (1) nprobe = nlist  ==  the flat index BIT FOR BIT after appends / overwrites / deletes, with plain and masked
filters; (2) a partial probe == the oracle's brute force restricted to (rows of the probed lists) U (delta rows), minus
tombstones; (3) save / load; (4) the same through HipIndexer / store_fhir_docs_in_opensearch with RASS_IVF_NLIST set
(automatic build at RASS_IVF_MIN_ROWS, rebuild past RASS_IVF_REBUILD_FRACTION)."""
import asyncio

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIM = 1024
TOL_F64 = 2e-6


def _clustered(rng, n, centres, sigma=1.0):
    lab = rng.integers(0, centres.shape[0], size=n)
    x = centres[lab] + sigma * rng.standard_normal((n, DIM)).astype(np.float32) / np.sqrt(DIM)
    return x.astype(np.float32)


@pytest.fixture(scope="module")
def world(gpu):
    from rassengine_amd.engine import Engine
    from rassengine_amd.ivf import IvfBackedIndex, IvfPolicy
    rng = np.random.default_rng(8)
    centres = rng.standard_normal((150, DIM)).astype(np.float32)
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    n0 = 20_011                                            # not a multiple of 32: 11 rows of the build land in the delta
    x0 = _clustered(rng, n0, centres)
    tags0 = (rng.integers(1, 5, size=n0) | (rng.integers(1, 3, size=n0) << 24)).astype(np.int32)
    eng = Engine(0, DIM)
    idx = IvfBackedIndex(eng.open_index("ivf-delta"), IvfPolicy.manual(nprobe=4))
    idx.add(x0, tags=tags0)
    for r in (3, 64, 19_999):
        idx.delete(r)                                      # tombstones BEFORE the build: never enter the IVF's slab
    idx.policy.iters = 6
    ivf = idx.build_ivf(nlist=64)
    assert ivf.covered_rows == n0 // 32 * 32 and idx.covered == 20_000 and idx.builds == 1
    # after the build: appends (the delta), deletes of covered rows, of delta rows, and an "overwrite" (delete + append)
    x1 = _clustered(rng, 3_000, centres)
    tags1 = (rng.integers(1, 5, size=3_000) | (rng.integers(1, 3, size=3_000) << 24)).astype(np.int32)
    first = idx.add(x1, tags=tags1)
    assert first == n0 and idx.covered == 20_000           # manual policy: no rebuild
    dead = [3, 64, 19_999, 100, 7_777, 20_005, n0 + 5, n0 + 2_999]
    for r in dead[3:]:
        idx.delete(r)
    x = np.concatenate([x0, x1])
    tags = np.concatenate([tags0, tags1])
    tags[dead] = -1
    q = (centres[rng.integers(0, 150, size=40)] + 0.7 * rng.standard_normal((40, DIM)).astype(np.float32) / np.sqrt(DIM)
         ).astype(np.float32)
    q[5] = x1[17]                                          # a query that IS a delta row
    q[6] = x0[100]                                         # ... and one that is a tombstoned covered row
    yield eng, idx, x, tags, q, dead
    eng.close()


def test_every_list_probed_plus_delta_equals_flat_bit_for_bit(world):
    from rassengine_amd.engine import FlatIndex
    eng, idx, x, tags, q, dead = world
    assert idx.rows == 23_011 and idx.count == 23_011 - len(dead) and idx.ivf.rows == 20_000 - 5
    PM, DM = 0x00FFFFFF, 0x7F000000
    for k, nq in ((10, 40), (1, 1), (32, 17), (5, 32)):
        s_f, i_f = FlatIndex.search(idx, q[:nq], k)                       # the flat scan of the same rows
        s_i, i_i = idx.search(q[:nq], k, nprobe=64)
        assert np.array_equal(i_i, i_f) and np.array_equal(s_i, s_f), (k, nq)
        assert not set(i_i.reshape(-1).tolist()) & set(dead)
    assert idx.search(q[5:6], 1, nprobe=64)[1][0, 0] == 20_011 + 17       # the delta row finds itself
    assert idx.search(q[6:7], 3, nprobe=64)[1][0, 0] != 100               # the tombstoned covered row is gone
    # plain (exact-tag) filters and masked ones (patient code / doc_type code), per query
    qf = np.array([(r % 4 + 1) | ((r % 2 + 1) << 24) for r in range(40)], dtype=np.int32)
    s_f, i_f = FlatIndex.search(idx, q, 10, qf)
    s_i, i_i = idx.search(q, 10, qf, nprobe=64)
    assert np.array_equal(i_i, i_f) and np.array_equal(s_i, s_f)
    for vals, mask in ((np.array([r % 4 + 1 for r in range(40)], dtype=np.int32), PM),
                       (np.array([(r % 2 + 1) << 24 for r in range(40)], dtype=np.int32), DM)):
        m = np.full(40, mask, dtype=np.int32)
        s_f, i_f = FlatIndex.search(idx, q, 10, vals, m)
        s_i, i_i = idx.search(q, 10, vals, m, nprobe=64)
        assert np.array_equal(i_i, i_f) and np.array_equal(s_i, s_f)
        live = i_i[i_i >= 0]
        assert live.size and np.all((tags[i_i[i_i >= 0]] & mask) == np.repeat(vals, 10)[(i_i >= 0).reshape(-1)])
    # k > 32 and exact=True are the flat index's answers
    s_f, i_f = FlatIndex.search(idx, q[:3], 50)
    s_i, i_i = idx.search(q[:3], 50)
    assert np.array_equal(i_i, i_f) and np.array_equal(s_i, s_f)
    assert np.array_equal(idx.search(q[:3], 10, exact=True)[1], FlatIndex.search(idx, q[:3], 10)[1])


def test_partial_probe_is_brute_force_over_probed_lists_and_delta(world, oracle):
    eng, idx, x, tags, q, dead = world
    ivf = idx.ivf
    covered, nlist, k = ivf.covered_rows, ivf.nlist, 10
    xn = idx.get_rows(0, idx.rows)                                         # the stored (normalised) rows
    qn = oracle.normalize_ref(q).astype(np.float32)
    assign = ivf.assign
    alive = tags != -1
    cn = oracle.normalize_ref(ivf.centroids.cpu().numpy()).astype(np.float32)
    coarse = oracle.scores(cn, qn, kind=oracle.KIND_F64)                   # [nq, nlist]: what the coarse scan ranks
    order = np.argsort(-coarse, axis=1, kind="stable")
    for nprobe in (1, 4, 16):
        s_g, i_g = idx.search(q, k, nprobe=nprobe)
        for r in range(q.shape[0]):
            got = i_g[r][i_g[r] >= 0]
            cov = got[got < covered]
            lists = set(assign[cov].tolist())
            assert len(lists) <= nprobe                                    # hits come from <= nprobe lists + the delta
            assert lists <= set(order[r, :nprobe + 1].tolist())            # ... the query's BEST lists (+1: a coarse near-tie)
            best = set(order[r, :nprobe].tolist())
            # the probed set: the nprobe best lists — or, on a coarse near-tie, the boundary list's neighbour instead
            probed = best if lists <= best else set(order[r, :nprobe - 1].tolist()) | {int(order[r, nprobe])}
            lists = probed
            # brute force restricted to (those lists) U (delta): the engine's hits must be its top-|got|... and every
            # row of the restricted set that beats the engine's last hit must be IN the result
            member = alive.copy()
            member[:covered] &= np.isin(assign, sorted(lists))
            rows = np.nonzero(member)[0]
            rs, ri = oracle.search(xn[rows], qn[r][None, :], k, kind=oracle.KIND_F64)
            want = np.where(ri[0] >= 0, rows[np.clip(ri[0], 0, None)], -1)
            if not np.array_equal(i_g[r], want):                           # fp32 near-ties may swap neighbours
                assert sorted(i_g[r].tolist()) == sorted(want.tolist()), (nprobe, r, i_g[r], want)
            valid = want >= 0
            assert np.all(np.abs(np.sort(s_g[r][valid].astype(np.float64)) - np.sort(rs[0][valid])) <= 2 * TOL_F64)
        # the delta is ALWAYS scanned: a query that is a delta row finds it at any nprobe
        assert i_g[5, 0] == 20_011 + 17
    # and the probe really is partial: the fine scans touch the probed lists + the delta, not the shard
    _, _, scanned1 = ivf.search_delta(idx, q, k, 1)
    _, _, scanned_all = ivf.search_delta(idx, q, k, 64)
    delta = idx.rows - covered
    # 40 queries = 2 batches, each touching every slab row (the 3 rows dead BEFORE the build never entered the slab; the 2
    # tombstoned since still occupy their slots, skipped by tag) + the delta
    assert scanned_all == 2 * (covered - 3 + delta)
    assert 2 * delta < scanned1 < 0.7 * scanned_all, (scanned1, scanned_all)


def test_device_path_and_save_load(world, tmp_path):
    import torch
    from rassengine_amd.engine import FlatIndex
    from rassengine_amd.ivf import IvfBackedIndex, IvfPolicy
    eng, idx, x, tags, q, dead = world
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(q[:32]).to(dev)
    out_s = torch.empty((32, 10), device=dev)
    out_i = torch.empty((32, 10), dtype=torch.int64, device=dev)
    idx.policy.nprobe = 64
    idx.search_device(qd.data_ptr(), 32, 10, out_s.data_ptr(), out_i.data_ptr())
    eng.synchronize()
    s_f, i_f = FlatIndex.search(idx, q[:32], 10)
    assert np.array_equal(out_i.cpu().numpy(), i_f) and np.array_equal(out_s.cpu().numpy(), s_f)
    idx.policy.nprobe = 4
    path = str(tmp_path / "backed.rass.tmp")                 # docstore hands a temporary name and renames it
    idx.save(path)
    import os
    os.replace(path, path[:-4])
    assert os.path.exists(str(tmp_path / "backed.rass.ivf")) and idx.saved_files(path[:-4]) == [str(tmp_path / "backed.rass.ivf")]
    back = IvfBackedIndex.load(eng, "ivf-delta-restored", path[:-4], IvfPolicy.manual(nprobe=4))
    try:
        assert back.rows == idx.rows and back.count == idx.count and back.covered == idx.covered
        for nprobe in (2, 64):
            a, b = idx.search(q, 10, nprobe=nprobe), back.search(q, 10, nprobe=nprobe)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
        back.delete(200)                                      # the loaded IVF still finds the row's slab position
        assert 200 not in back.search(x[200:201], 5, nprobe=64)[1][0].tolist()
        assert 200 in idx.search(x[200:201], 5, nprobe=64)[1][0].tolist()
    finally:
        eng.drop_index("ivf-delta-restored")


def test_through_the_boundary_with_automatic_builds(gpu, monkeypatch):
    """RASS_IVF_NLIST on: store_fhir_docs_in_opensearch -> the index builds its IVF at RASS_IVF_MIN_ROWS, keeps a flat
    delta, rebuilds past the threshold; HipIndexer answers equal the flat engine's (nprobe = nlist) after overwrites."""
    from rassengine_amd import config, embedding, indexer
    from rassengine_amd.docstore import REGISTRY
    from rassengine_amd.engine import Engine, FlatIndex
    from rassengine_amd.ivf import IvfBackedIndex
    from tests.helpers import HashEmbedder
    monkeypatch.setattr(config, "RASS_IVF_NLIST", 16)
    monkeypatch.setattr(config, "RASS_IVF_NPROBE", 16)
    monkeypatch.setattr(config, "RASS_IVF_MIN_ROWS", 1500)
    monkeypatch.setattr(config, "RASS_IVF_REBUILD_FRACTION", 0.5)
    monkeypatch.setattr(config, "RASS_KNN_PREFETCH", 0)
    REGISTRY.clear()
    REGISTRY.set_index_factory(None)                          # the default factory: the process-global engine
    embedding.set_embedder(HashEmbedder(1024))
    name = "rass-idx-ivf-user"
    try:
        docs = [{"doc_id": f"n-{i}", "doc_type": "unstructured", "patientId": f"p{i % 3}",
                 "unstructuredText": f"note {i} topic{i % 13} drug{i % 7} ward{i % 5}"} for i in range(3200)]
        builds = []
        for a in range(0, 3200, 400):
            asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs[a:a + 400], None, name))
            st = REGISTRY.get(name)
            builds.append((st.index.rows, st.index.builds, st.index.covered))
        idx = REGISTRY.get(name).index
        assert isinstance(idx, IvfBackedIndex)
        assert [b[1] for b in builds] == [0, 0, 0, 1, 1, 1, 2, 2], builds     # built at 1600 rows, rebuilt once the delta > 800
        assert builds[3][2] == 1600 and builds[6][2] == 2784 and idx.covered == 2784 and idx.rows == 3200
        # overwrites: tombstone covered rows and delta rows, append the new versions
        asyncio.run(indexer.store_fhir_docs_in_opensearch(
            [], [dict(docs[7], unstructuredText="entirely new words here"), dict(docs[3100], unstructuredText="other words")],
            None, name))
        ix = indexer.HipIndexer(None, name)
        for text, kw in (("note 12 topic12 drug5 ward2", {}), ("entirely new words here", {}),
                         ("note 300 topic1 drug6 ward0", {"patient_id": "p0"}), ("other words", {"patient_id": "p1"})):
            q = asyncio.run(embedding.embed_query(text))
            got = ix.semantic_search(q, k=8, **kw)
            pid = kw.get("patient_id")
            st = REGISTRY.get(name)
            flt = st.filter_for(pid, None)
            f = m = None
            if flt[1]:
                f, m = np.array([flt[0]], dtype=np.int32), np.array([flt[1]], dtype=np.int32)
            s_f, i_f = FlatIndex.search(idx, q, 8, f, m)
            want = [(st.row_doc[int(r)]["doc_id"]) for r in i_f[0] if r >= 0]
            assert [d["doc_id"] for d, _ in got] == want, (text, kw)
            assert np.allclose([s for _, s in got], 1.0 / (2.0 - s_f[0][:len(want)]), rtol=0, atol=1e-6)
        assert ix.semantic_search(asyncio.run(embedding.embed_query("entirely new words here")), k=1)[0][0]["doc_id"] == "n-7"
        assert idx.count == 3200 and idx.rows == 3202
    finally:
        embedding.set_embedder(None)
        REGISTRY.clear()
        try:
            Engine.get(config.RASS_DEVICE, config.EMBED_DIM).drop_index(name)
        except Exception:
            pass
