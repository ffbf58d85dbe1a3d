"""GPU tests of the IVF over a bf16 slab (rass_ivf_build_ex(..., RASS_BF16); SURVEY §8f-4's bf16 path for cfg 5).

What is pinned: a probe returns what a FLAT bf16 index over the same rows returns, restricted to the probed lists —
 * nprobe = nlist reproduces the flat bf16 index bit for bit (same kernel arithmetic: bf16 rows, bf16-rounded queries,
   fp32 accumulation, the K slices added in the same order), through the top-k path (nlist <= 32) and the threshold path;
 * a partial probe equals the oracle's brute force over the bf16-rounded operands restricted to the rows of the lists whose
   centroids score best for the query (the coarse scan is the exact fp32 one, as for the fp32 IVF), ids identical up to
   ties, scores within 2e-6 of fp64; `scanned` = the rows of the union of a batch's probed lists, the same number the fp32
   IVF over the same centroids reports;
 * filters, tombstones, save / load (the file keeps the dtype), and what is refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 2e-6


def _bf16_round(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().float().numpy()


def _clustered(rng, n, dim, centres, sigma):
    c = rng.standard_normal((centres, dim)).astype(np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    lab = rng.integers(0, centres, size=n)
    x = c[lab] + sigma * rng.standard_normal((n, dim)).astype(np.float32) / np.sqrt(dim)
    return x.astype(np.float32), c


DEAD = (5, 77, 12345, 29999)


@pytest.fixture(scope="module")
def built(gpu):
    from rassengine_amd.engine import Engine
    from rassengine_amd.ivf import IvfIndex, train_centroids
    rng = np.random.default_rng(4242)
    n, dim = 30000, 1024
    x, centres = _clustered(rng, n, dim, 200, 1.0)
    tags = rng.integers(1, 5, size=n).astype(np.int32)
    eng = Engine(0, dim)
    flat = eng.open_index("ivfb-src")
    flat.add(x, tags=tags)
    flat_b = eng.open_index("ivfb-flat-bf16", dtype="bf16")
    flat_b.add(x, tags=tags)
    for r in DEAD:
        flat.delete(r)
        flat_b.delete(r)
    cent = train_centroids(flat, 128, train_rows=0, iters=8, seed=3)
    ivf_b = IvfIndex.build(flat, nlist=128, centroids=cent, dtype="bf16")
    ivf_f = IvfIndex.build(flat, nlist=128, centroids=cent)
    q = centres[rng.integers(0, 200, size=50)] + 0.8 * rng.standard_normal((50, dim)).astype(np.float32) / np.sqrt(dim)
    yield eng, flat, flat_b, ivf_b, ivf_f, cent.cpu().numpy(), q.astype(np.float32), tags
    ivf_b.close()
    ivf_f.close()
    eng.close()


def test_probe_all_lists_equals_the_flat_bf16_index(built):
    eng, flat, flat_b, ivf_b, ivf_f, cent, q, tags = built
    from rassengine_amd.ivf import IvfIndex
    assert ivf_b.dtype == "bf16" and ivf_f.dtype == "f32"
    assert ivf_b.nlist == 128 and ivf_b.rows == flat.count == flat_b.count
    # the slab holds bf16(the fp32 index's rows) = the flat bf16 index's rows
    assert np.array_equal(flat_b.get_rows(0, 64), _bf16_round(flat.get_rows(0, 64)))
    s_f, i_f = flat_b.search(q, 10)
    # threshold path (nprobe > 32), every list
    s_all, i_all, scanned = ivf_b.search(q, 10, nprobe=128)
    assert np.array_equal(i_all, i_f) and np.array_equal(s_all, s_f)
    assert scanned == 2 * flat.count
    # top-k path with nlist <= 32
    ivf32 = IvfIndex.build(flat, nlist=32, iters=5, seed=1, dtype="bf16")
    try:
        s_i, i_i, scanned = ivf32.search(q, 10, nprobe=32)
        assert np.array_equal(i_i, i_f) and np.array_equal(s_i, s_f)
        assert scanned == 2 * flat.count
        # one query, 17 queries (the 16-query kernel variant and a ragged second half)
        for nq in (1, 17):
            s1, i1, _ = ivf32.search(q[:nq], 7, nprobe=32)
            s2, i2 = flat_b.search(q[:nq], 7)
            assert np.array_equal(i1, i2) and np.array_equal(s1, s2)
    finally:
        ivf32.close()


@pytest.mark.parametrize("nprobe", [1, 4, 32, 33, 64])
def test_partial_probe_is_restricted_brute_force_on_bf16_operands(built, oracle, nprobe):
    eng, flat, flat_b, ivf_b, ivf_f, cent, q, tags = built
    import torch
    from rassengine_amd import ops
    k = 10
    n = flat.rows
    stored = flat_b.get_rows(0, n)                          # the bf16 rows, exactly
    qn_gpu = ops.normalize_rows(torch.from_numpy(q).cuda()).cpu().numpy()
    qb = _bf16_round(qn_gpu)
    all64 = oracle.scores(stored, qb)                       # fp64 truth over the bf16 operands
    assign = ivf_b.assign
    assert np.array_equal(assign, ivf_f.assign)
    deleted = np.zeros(n, dtype=bool)
    deleted[list(DEAD)] = True
    live_len = np.bincount(assign[~deleted], minlength=128)
    cn = oracle.normalize_ref(cent).astype(np.float32)
    qn = oracle.normalize_ref(q).astype(np.float32)
    coarse = oracle.scores(cn, qn, kind=oracle.KIND_F64)
    order = np.argsort(-coarse, axis=1, kind="stable")

    s_g, i_g, scanned = ivf_b.search(q, k, nprobe=nprobe)
    s_32, i_32, scanned_32 = ivf_f.search(q, k, nprobe=nprobe)
    assert scanned == scanned_32                            # same centroids, same coarse scan: the same lists are probed

    ambiguous = 0
    expect_scanned, union_exact = 0, True
    for b0 in range(0, q.shape[0], 32):
        union = set()
        for r in range(b0, min(b0 + 32, q.shape[0])):
            lists = order[r, :nprobe]
            gap = coarse[r, order[r, nprobe - 1]] - coarse[r, order[r, nprobe]]
            candidates = [lists]
            if gap < 1e-6:
                ambiguous += 1
                union_exact = False
                candidates.append(np.concatenate([order[r, :nprobe - 1], order[r, nprobe:nprobe + 1]]))
                candidates.append(order[r, :nprobe + 1])
            ok = False
            for cand in candidates:
                rows = np.nonzero(np.isin(assign, cand) & ~deleted)[0]
                sc = all64[r, rows]
                top = rows[np.lexsort((rows, -sc))[:k]]
                got = i_g[r]
                if len(top) < k:
                    assert np.all(got[len(top):] == -1) and np.all(np.isneginf(s_g[r][len(top):]))
                    got = got[:len(top)]
                # ids equal up to swaps / boundary replacements between rows whose fp64 scores tie within 2 TOL
                same = all(a == b or abs(all64[r, a] - all64[r, b]) <= 2 * TOL for a, b in zip(got, top))
                if same and set(got.tolist()) <= set(rows.tolist()):
                    assert np.all(np.abs(s_g[r][:len(got)].astype(np.float64) - all64[r, got]) <= TOL)
                    ok = True
                    break
            assert ok, (nprobe, r, i_g[r])
            union.update(int(l) for l in lists)
        expect_scanned += int(live_len[sorted(union)].sum())
    assert ambiguous <= 2
    if union_exact:
        assert scanned == expect_scanned
    # against the fp32 IVF over the same lists: the same neighbours up to bf16 rounding
    recall = np.mean([len(set(i_g[r]) & set(i_32[r])) / k for r in range(q.shape[0])])
    assert recall >= 0.9, recall
    both = i_g == i_32
    assert np.abs(s_g[both] - s_32[both]).max() <= 2e-3


def test_filters_and_tombstones(built):
    eng, flat, flat_b, ivf_b, ivf_f, cent, q, tags = built
    qf = np.array([(r % 4) + 1 for r in range(q.shape[0])], dtype=np.int32)
    s, i, _ = ivf_b.search(q, 10, nprobe=16, q_filter=qf)
    for r in range(q.shape[0]):
        live = i[r][i[r] >= 0]
        assert np.all(tags[live] == qf[r])
        assert not set(live.tolist()) & set(DEAD)
    # every list probed + filter = the flat bf16 index with the same filter
    s_a, i_a, _ = ivf_b.search(q, 10, nprobe=128, q_filter=qf)
    s_f, i_f = flat_b.search(q, 10, q_filter=qf)
    assert np.array_equal(i_a, i_f) and np.array_equal(s_a, s_f)


def test_save_load_keeps_the_dtype_and_the_bits(built, tmp_path):
    from rassengine_amd.ivf import IvfIndex
    from rassengine_amd._native import RassError
    eng, flat, flat_b, ivf_b, ivf_f, cent, q, tags = built
    path = str(tmp_path / "shard0.bf16.ivf")
    ivf_b.save(path)
    back = IvfIndex.load(eng, path)
    try:
        assert back.dtype == "bf16" and back.nlist == 128 and back.rows == ivf_b.rows
        for nprobe in (1, 8, 40, 128):
            s0, i0, sc0 = ivf_b.search(q, 10, nprobe)
            s1, i1, sc1 = back.search(q, 10, nprobe)
            assert np.array_equal(i0, i1) and np.array_equal(s0, s1) and sc0 == sc1
    finally:
        back.close()
    import os
    # the fp32 file of the same lists is about twice as long; a truncated bf16 file is refused
    p32 = str(tmp_path / "shard0.f32.ivf")
    ivf_f.save(p32)
    assert 1.8 < os.path.getsize(p32) / os.path.getsize(path) < 2.1
    raw = open(path, "rb").read()
    bad = str(tmp_path / "short.ivf")
    open(bad, "wb").write(raw[:-4096])
    with pytest.raises(RassError):
        IvfIndex.load(eng, bad)


def test_what_a_bf16_slab_cannot_do_is_refused(gpu):
    from rassengine_amd.engine import Engine
    from rassengine_amd.ivf import IvfIndex
    from rassengine_amd._native import RassError
    rng = np.random.default_rng(0)
    eng = Engine(0, 384)                       # row stride 384: not a multiple of the bf16 scan's 256-column K split
    try:
        flat = eng.open_index("odd")
        flat.add(rng.standard_normal((500, 384)).astype(np.float32))
        ok = IvfIndex.build(flat, nlist=8, iters=2, seed=0)          # the fp32 slab serves it
        ok.close()
        with pytest.raises(RassError):
            IvfIndex.build(flat, nlist=8, iters=2, seed=0, dtype="bf16")
        with pytest.raises(ValueError):
            IvfIndex.build(flat, nlist=8, iters=2, seed=0, dtype="fp8")
    finally:
        eng.close()
