"""Prefilter mode 2 (SURVEY §8f-4 "bf16 (or int8)"): int8 candidate scan + exact fp32 re-rank.

The candidate scan is integer work: its lists (scores AND rows) must equal the oracle's numpy restatement BIT FOR BIT.
The final result must be bit-identical to the flat fp32 path for every returned row; on these corpora the id lists
are identical too (recall 1.0 — the true top-k sits inside the int8 top-32)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def engine(gpu):
    from rassengine_amd.engine import Engine
    eng = Engine(device=0, dim=1024)
    yield eng
    eng.close()


def _cand(idx, q, qf=None):
    import torch
    s, r = idx.candidates_device(torch.from_numpy(np.ascontiguousarray(q)).cuda(), qf)
    return s.cpu().numpy(), r.cpu().numpy()


@pytest.mark.parametrize("dim", [1024, 384, 512, 768])
def test_int8_candidates_equal_the_oracle_bit_for_bit(gpu, oracle, dim):
    from rassengine_amd.engine import Engine
    eng = Engine(0, dim)
    try:
        rng = np.random.default_rng(100 + dim)
        n = 5000 + 37                                   # a ragged last tile and a ragged last 16-row block
        x = rng.standard_normal((n, dim)).astype(np.float32)
        x[11] = 0.0                                     # a zero row (a blank chunk's embedding): scale 0, score 0
        x[12, 5] = 40.0                                 # an outlier component: a coarse scale for that row
        tags = rng.integers(0, 3, size=n).astype(np.int32)
        idx = eng.open_index("i8")
        idx.set_prefilter("int8")                       # before any row exists
        assert idx.prefilter_mode == "int8"
        idx.add(x[:3000], tags=tags[:3000])
        idx.add(x[3000:], tags=tags[3000:])             # growth + an append that re-quantises a partly filled block
        idx.delete(17)
        idx.delete(4999)
        tags[[17, 4999]] = -1
        xn = idx.get_rows(0, n)                         # the stored (normalised) rows: what the quantiser saw
        for nq in (1, 16, 17, 32):
            q = rng.standard_normal((nq, dim)).astype(np.float32)
            qn = oracle.normalize_c(q)
            qf = rng.integers(-1, 3, size=nq).astype(np.int32)
            for f in (None, qf):
                s_g, r_g = _cand(idx, q, f)
                s_o, r_o = oracle.candidates_i8(xn, qn, 32, tags=tags, qfilter=f)
                assert np.array_equal(r_g, r_o), (dim, nq, f is not None)
                assert np.array_equal(s_g.view(np.uint32), s_o.view(np.uint32)), (dim, nq)
    finally:
        eng.close()


def test_int8_prefilter_matches_flat_path(engine, oracle):
    rng = np.random.default_rng(8)
    idx = engine.open_index("pf8")
    idx.set_prefilter("int8")
    n = 0
    tags_all = []
    for c in (1, 40, 3000, 17, 9000):            # odd batch sizes: int8 slab + scales kept in sync on add + growth
        x = rng.standard_normal((c, 1024), dtype=np.float32)
        t = rng.integers(1, 4, size=c).astype(np.int32)
        idx.add(x, tags=t)
        tags_all.append(t)
        n += c
    tags = np.concatenate(tags_all)
    idx.delete(7)
    idx.delete(4000)
    q = rng.standard_normal((45, 1024), dtype=np.float32)
    qf = rng.integers(-1, 4, size=45).astype(np.int32)
    for k in (1, 5, 10, 16, 32):                 # k > 16 silently takes the exact flat scan
        s_p, i_p = idx.search(q, k, q_filter=qf)
        idx.set_prefilter(False)
        s_f, i_f = idx.search(q, k, q_filter=qf)
        idx.set_prefilter("int8")                # re-enabled on a populated index: quantises the existing rows
        assert np.array_equal(i_p, i_f), k
        assert np.array_equal(s_p, s_f), k       # exact re-rank = the flat kernel's fmaf order
    live = i_p[i_p >= 0]
    assert 7 not in live and 4000 not in live
    for r in range(45):
        if qf[r] >= 0:
            assert np.all(tags[i_p[r][i_p[r] >= 0]] == qf[r])
    # the two candidate copies are exclusive; switching rebuilds from the fp32 rows
    idx.set_prefilter("bf16")
    assert idx.prefilter_mode == "bf16"
    s_b, i_b = idx.search(q, 10, q_filter=qf)
    idx.set_prefilter("int8")
    s_8, i_8 = idx.search(q, 10, q_filter=qf)
    assert np.array_equal(i_b, i_8) and np.array_equal(s_b, s_8)


def test_int8_prefilter_small_and_padding(engine):
    rng = np.random.default_rng(9)
    idx = engine.open_index("pf8-small")
    x = rng.standard_normal((5, 1024), dtype=np.float32)
    idx.add(x)
    idx.set_prefilter("int8")
    s, i = idx.search(x[:2], 10)
    idx.set_prefilter(False)
    s2, i2 = idx.search(x[:2], 10)
    assert np.array_equal(i, i2) and np.array_equal(s, s2)
    assert np.all(i[:, 5:] == -1) and np.all(np.isneginf(s[:, 5:]))
    assert i[0, 0] == 0 and i[1, 0] == 1


def test_int8_near_duplicates_are_reranked_exactly(engine):
    """Rows closer together than the int8 resolution: the candidate scan cannot order them, the fp32 re-rank must."""
    rng = np.random.default_rng(10)
    base = rng.standard_normal((1, 1024)).astype(np.float32)
    x = np.repeat(base, 20, axis=0) + 1e-4 * rng.standard_normal((20, 1024)).astype(np.float32)
    x[11] = x[3]                                  # exact duplicate
    filler = rng.standard_normal((2000, 1024)).astype(np.float32)
    idx = engine.open_index("pf8-dup")
    idx.add(np.concatenate([filler, x]))
    q = base * 3.0
    for k in (12, 16):
        idx.set_prefilter("int8")
        s_p, i_p = idx.search(q, k)
        idx.set_prefilter(False)
        s_f, i_f = idx.search(q, k)
        assert np.array_equal(i_p, i_f) and np.array_equal(s_p, s_f), k


def test_int8_recall_on_a_large_clustered_corpus(engine, oracle):
    """200 k rows in 2 000 tight clusters (the hard case for a coarse candidate score: many near neighbours): recall@10
    of the int8 prefilter against the flat scan of the same index."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(5)
    centres = torch.randn((2000, 1024), generator=g, device="cuda")
    centres /= centres.norm(dim=1, keepdim=True)
    idx = engine.open_index("pf8-big", capacity_rows=200_000)
    for c0 in range(0, 200_000, 50_000):
        which = torch.randint(0, 2000, (50_000,), generator=g, device="cuda")
        rows = centres[which] + 0.02 * torch.randn((50_000, 1024), generator=g, device="cuda")
        idx.add(rows.cpu().numpy())
    q = (centres[:64] + 0.02 * torch.randn((64, 1024), generator=g, device="cuda")).cpu().numpy()
    s_f, i_f = idx.search(q, 10)
    idx.set_prefilter("int8")
    s_p, i_p = idx.search(q, 10)
    recall = np.mean([len(set(i_p[r]) & set(i_f[r])) / 10 for r in range(64)])
    assert recall >= 0.95, recall
    same = i_p == i_f
    assert np.array_equal(s_p[same], s_f[same])            # every returned row carries its exact fp32 score


def test_mode_validation(gpu):
    from rassengine_amd.engine import Engine
    from rassengine_amd._native import RassError
    eng = Engine(0, 1536)
    try:
        idx = eng.open_index("wide")
        with pytest.raises(RassError):
            idx.set_prefilter("int8")               # wide rows: fp32 flat scan only
        with pytest.raises(ValueError):
            idx.set_prefilter("fp4")
    finally:
        eng.close()


@pytest.mark.parametrize("mode", ["int8", "bf16"])
def test_prefilter_batch_call_equals_groups(gpu, mode):
    """rass_index_search_device_batch on an index in a prefilter mode (one normalise, one query conversion, the groups'
    candidate scans, ONE grouped merge, ONE re-rank launch) ≡ the group-by-group calls, bit for bit — ragged last group,
    per-query filters, tombstones, an id base, k above the prefilter's limit (exact flat batch)."""
    from rassengine_amd.engine import Engine
    torch = gpu
    rng = np.random.default_rng(21)
    n = 30_011
    x = rng.standard_normal((n, 1024)).astype(np.float32)
    tags = rng.integers(1, 20, size=n).astype(np.int32)
    eng = Engine(0, 1024)
    try:
        eng.set_stream(int(torch.cuda.current_stream().cuda_stream))
        ix = eng.open_index("pfb")
        ix.add(x, tags=tags)
        ix.delete(5)
        ix.delete(29_000)
        ix.set_prefilter(mode)
        g = torch.Generator(device="cuda"); g.manual_seed(9)
        for nq, k in ((33, 10), (100, 7), (256, 16), (70, 20)):
            q = torch.randn((nq, 1024), generator=g, device="cuda")
            filt = torch.randint(1, 20, (nq,), dtype=torch.int32, device="cuda")
            filt[1] = -1
            for f in (None, filt):
                outs = []
                for batch in (True, False):
                    s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
                    i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
                    if batch:
                        ix.search_device_batch(q.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), id_base=1_000_000,
                                               d_q_filter_ptr=f.data_ptr() if f is not None else 0)
                    else:
                        for o in range(0, nq, 32):
                            b = min(32, nq - o)
                            ix.search_device(q[o:o + b].data_ptr(), b, k, s[o:o + b].data_ptr(), i[o:o + b].data_ptr(),
                                             id_base=1_000_000, d_q_filter_ptr=f[o:o + b].data_ptr() if f is not None else 0)
                    torch.cuda.synchronize()
                    outs.append((s.cpu().numpy(), i.cpu().numpy()))
                assert np.array_equal(outs[0][1], outs[1][1]), (mode, nq, k)
                assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32)), (mode, nq, k)
        # against the flat scan of the same index: identical on this well-separated corpus
        q = torch.randn((64, 1024), generator=g, device="cuda")
        s_p = torch.empty((64, 10), dtype=torch.float32, device="cuda"); i_p = torch.empty((64, 10), dtype=torch.int64, device="cuda")
        ix.search_device_batch(q.data_ptr(), 64, 10, s_p.data_ptr(), i_p.data_ptr())
        torch.cuda.synchronize()
        ix.set_prefilter(False)
        s_f = torch.empty_like(s_p); i_f = torch.empty_like(i_p)
        ix.search_device_batch(q.data_ptr(), 64, 10, s_f.data_ptr(), i_f.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(i_p, i_f) and torch.equal(s_p, s_f)
    finally:
        eng.close()


def test_int8_candidates_with_the_sample_floor(gpu, oracle, monkeypatch):
    """A slab long enough for the sample floor (>= 8 samples of 64 rows x 256 workgroups): the candidate lists with the floor
    ≡ without it ≡ the oracle, bit for bit — plain, filtered (a filter matching few rows: fewer than 32 finite sample maxima,
    no floor for that query) and with tombstones inside the sample."""
    from rassengine_amd.engine import Engine
    torch = gpu
    dim, n = 512, 150_000
    eng = Engine(0, dim)
    try:
        g = torch.Generator(device="cuda").manual_seed(77)
        idx = eng.open_index("i8-floor", capacity_rows=n)
        tags = np.zeros(n, dtype=np.int32)
        rng = np.random.default_rng(1)
        tags[:] = rng.integers(1, 4, size=n)
        tags[rng.choice(n, 40, replace=False)] = 9          # a rare tag
        for c0 in range(0, n, 50_000):
            x = torch.randn((50_000, dim), generator=g, device="cuda").cpu().numpy()
            idx.add(x, tags=tags[c0:c0 + 50_000])
        for r in (3, 64, 9_000, 149_999):
            idx.delete(r)
            tags[r] = -1
        idx.set_prefilter("int8")
        xn = idx.get_rows(0, n)
        q = torch.randn((32, dim), generator=g, device="cuda").cpu().numpy()
        qn = oracle.normalize_c(q)
        qf = rng.integers(-1, 4, size=32).astype(np.int32)
        qf[5] = 9
        qf[6] = 77                                          # matches nothing
        for f in (None, qf):
            s_o, r_o = oracle.candidates_i8(xn, qn, 32, tags=tags, qfilter=f)
            for floor in ("1", "0"):
                monkeypatch.setenv("RASS_I8_SAMPLE_FLOOR", floor)
                s_g, r_g = _cand(idx, q, f)
                assert np.array_equal(r_g, r_o), (floor, f is not None)
                assert np.array_equal(s_g.view(np.uint32), s_o.view(np.uint32)), floor
        assert np.all(r_o[6] == -1) if qf is not None else True
    finally:
        eng.close()


@pytest.mark.parametrize("mode", ["int8", "bf16"])
def test_prefilter_knob_behind_the_boundary(gpu, monkeypatch, mode):
    """RASS_PREFILTER: the default index factory opens every index in that mode; an ask()-shaped flow through
    HipIndexer returns what the exact index returns (docs, order and the reference's 1 / (2 - cos) scores)."""
    import asyncio
    from rassengine_amd import config, embedding, indexer
    from rassengine_amd.docstore import REGISTRY
    from rassengine_amd.engine import Engine
    from tests.helpers import HashEmbedder

    docs = [{"doc_id": f"text-f-{i}", "doc_type": "unstructured", "patientId": f"p{i % 3}",
             "unstructuredText": f"note {i} mentions condition{i % 11} and drug{i % 5}"} for i in range(400)]
    results = {}
    for setting in ("off", mode):
        monkeypatch.setattr(config, "RASS_PREFILTER", setting)
        REGISTRY.clear()
        REGISTRY.set_index_factory(None)
        embedding.set_embedder(HashEmbedder(1024))
        try:
            name = "rass-idx-knob"
            asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs, None, name))
            assert REGISTRY.get(name).index.prefilter_mode == setting
            q = asyncio.run(embedding.embed_query("note about condition7 and drug2"))
            ix = indexer.HipIndexer(None, name)
            results[setting] = (ix.semantic_search(q, k=5, patient_id="p1"), ix.semantic_search(q, k=10),
                                ix.semantic_search(q, k=40))          # k > 16: the exact scan either way
        finally:
            embedding.set_embedder(None)
            REGISTRY.clear()
            Engine.get(config.RASS_DEVICE, config.EMBED_DIM).drop_index(name)
    for a, b in zip(results["off"], results[mode]):
        assert [d["doc_id"] for d, _ in a] == [d["doc_id"] for d, _ in b]
        assert [s for _, s in a] == [s for _, s in b]


@pytest.mark.parametrize("mode", ["int8", "bf16"])
def test_prefilter_with_global_ids_and_masked_filters(engine, mode):
    """A shard of a multi-GPU index: caller-assigned global ids (reported, and the tie order) and masked tag compares
    (patient code | doc_type << 24) go through the candidate scan + re-rank like plain searches — ≡ the exact scan."""
    rng = np.random.default_rng(55)
    idx = engine.open_index("pf-gid-" + mode)
    n = 6000
    x = rng.standard_normal((n, 1024)).astype(np.float32)
    x[100] = x[50]                                           # an exact duplicate: equal scores, the lower GLOBAL id first
    tags = (rng.integers(1, 6, size=n) | (rng.integers(0, 2, size=n) << 24)).astype(np.int32)
    idx.add(x[:2500], tags=tags[:2500], first_global_id=10_000)
    idx.add(x[2500:], tags=tags[2500:], first_global_id=50_000)
    idx.delete(7)
    q = np.concatenate([x[50:51] * 2.0, rng.standard_normal((40, 1024)).astype(np.float32)])
    f = np.array([(r % 5) + 1 if r % 2 else 0x01000000 for r in range(41)], dtype=np.int32)
    m = np.array([0x00ffffff if r % 2 else 0x01000000 for r in range(41)], dtype=np.int32)
    f[0], m[0] = -1, -1
    for k in (1, 10, 16):
        idx.set_prefilter(mode)
        a = idx.search(q, k, q_filter=f, q_filter_mask=m)
        b = idx.search(q, k)
        idx.set_prefilter(False)
        a0 = idx.search(q, k, q_filter=f, q_filter_mask=m)
        b0 = idx.search(q, k)
        assert np.array_equal(a[1], a0[1]) and np.array_equal(a[0], a0[0]), k
        assert np.array_equal(b[1], b0[1]) and np.array_equal(b[0], b0[0]), k
    assert b0[1][0, 0] == 10_050 and (k == 1 or b0[1][0, 1] == 10_100)      # the duplicate pair in global-id order
    assert b0[1].min() >= 10_000
