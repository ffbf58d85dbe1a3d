"""GPU parity of the fused flat cosine scan + top-k (K1+K2) against the CPU oracle.

Bars (BASELINE.json north_star): top-k ids identical to the fp64 brute-force ranking
(score desc, id asc); cosine within 1e-3 — here the fp32 path is held to BIT-EQUALITY with
the oracle's emulation of the kernel's fmaf order, and to 2e-6 of the fp64 truth.
Every call goes through the C ABI (rass_scan_topk_f32 / rass_index_*).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_F64 = 2e-6  # |fp32 fmaf-chain score - fp64 score| for unit vectors, D <= 1024
TOL_F64_WIDE = 3e-6  # the same for 1 024 < D <= 2 048 (twice as many fmaf steps per slice)


def _stride(dim):
    """api.hip pad_stride: whole 128-column units up to 1 024 columns, whole 256-column units above."""
    return (dim + 127) // 128 * 128 if dim <= 1024 else (dim + 255) // 256 * 256


def _pad(x, stride):
    out = np.zeros((x.shape[0], stride), dtype=np.float32)
    out[:, : x.shape[1]] = x
    return out


def _ids_match_with_ties(ids_gpu, s_gpu, ids_ref, s_ref, all_scores64):
    """ids must equal the fp64 ranking except where fp64 scores of the swapped rows are
    closer than the fp32 rounding error."""
    if np.array_equal(ids_gpu, ids_ref):
        return True
    for q in range(ids_ref.shape[0]):
        for e in range(ids_ref.shape[1]):
            a, b = ids_gpu[q, e], ids_ref[q, e]
            if a == b:
                continue
            if a < 0 or b < 0:
                return False
            if abs(all_scores64[q, a] - all_scores64[q, b]) > 2 * TOL_F64:
                return False
    return True


def _run_scan(torch, xn, q_raw, k, dim, stride, tags=None, qfilter=None, id_base=0):
    from rassengine_amd import ops
    corpus = torch.from_numpy(np.ascontiguousarray(xn)).cuda()
    queries = torch.from_numpy(np.ascontiguousarray(q_raw, dtype=np.float32)).cuda()
    t = None if tags is None else torch.from_numpy(tags.astype(np.int32)).cuda()
    f = None if qfilter is None else torch.from_numpy(qfilter.astype(np.int32)).cuda()
    s, i = ops.scan_topk(corpus, queries, k, row_tag=t, q_filter=f, id_base=id_base)
    torch.cuda.synchronize()
    return s.cpu().numpy(), i.cpu().numpy()


@pytest.mark.parametrize("n,dim,nq,k", [
    (1, 1024, 1, 1),
    (31, 1024, 3, 5),
    (32, 1024, 16, 10),
    (33, 1024, 17, 10),
    (1000, 1024, 32, 32),
    (4096 + 17, 1024, 5, 3),
    (20000, 1024, 32, 10),
    (3000, 384, 8, 10),
    (3000, 100, 4, 5),     # dim padded to 128
    (2500, 768, 20, 10),
    (3000, 512, 32, 10),   # CH = 4
    (3000, 640, 32, 10),   # CH = 5
    (2000, 896, 7, 5),     # CH = 7
    (2000, 600, 32, 8),    # dim padded to 640
    (9000, 500, 9, 4),     # dim padded to 512
    (5, 1024, 2, 10),      # fewer rows than k
    (45000, 256, 8, 10),   # full grid, B <= 16: XCD-skewed tile order, 1.2 super-rounds
    (100003, 128, 16, 7),  # ... 2.7 super-rounds, ragged last tile
    (70000, 128, 1, 32),
    # wide rows (1 024 < dim <= 2 048): the panel kernel, 16 queries per launch, groups of 17..32 split
    (3000, 1536, 16, 10),  # 2 panels x 6 chunks
    (3000, 2048, 32, 10),  # 4 panels x 4 chunks; 32 queries = two launches
    (2000, 1100, 5, 5),    # padded to 1280: 2 x 5
    (1500, 1792, 20, 7),   # 2 x 7
    (33, 2048, 17, 32),
    # the other two panel variants: stride 1 280 = <5, 2> and 1 792 = <7, 2> (the closest to the spill limit)
    (3000, 1100, 16, 10), (2500, 1280, 32, 32), (777, 1250, 17, 5), (40000, 1280, 1, 10),
    (3000, 1600, 16, 10), (2500, 1792, 32, 32), (777, 1700, 17, 5), (40000, 1792, 16, 32),
    (40000, 2048, 16, 10),  # full grid, XCD-skewed tile order, ragged last tile
    (1, 1536, 1, 1),
])
def test_scan_matches_oracle(gpu, oracle, n, dim, nq, k):
    rng = np.random.default_rng(1000 + n + dim + nq + k)
    xn = oracle.normalize_ref(rng.standard_normal((n, dim), dtype=np.float32)).astype(np.float32)
    q_raw = rng.standard_normal((nq, dim), dtype=np.float32) * 3.0  # un-normalised on purpose
    qn = oracle.normalize_c(q_raw)
    stride = _stride(dim)
    tol = TOL_F64 if dim <= 1024 else TOL_F64_WIDE

    s_gpu, i_gpu = _run_scan(gpu, xn, q_raw, k, dim, stride)

    s64, i64 = oracle.search(xn, qn, k, kind=oracle.KIND_F64)
    all64 = oracle.scores(xn, qn, kind=oracle.KIND_F64)
    assert _ids_match_with_ties(i_gpu, s_gpu, i64, s64, all64), (i_gpu, i64)
    valid = i64 >= 0
    assert np.array_equal(i_gpu >= 0, valid)
    assert np.all(np.abs(s_gpu[valid].astype(np.float64) - s64[valid]) <= tol)
    assert np.all(np.isneginf(s_gpu[~valid]))

    # bit-equality with the emulated fmaf order — needs the GPU-normalised queries, which may
    # differ from numpy's by an ulp, so emulate with the queries the GPU actually used
    from rassengine_amd import ops
    qn_gpu = ops.normalize_rows(gpu.from_numpy(q_raw).cuda()).cpu().numpy()
    s32, i32 = oracle.search(xn, qn_gpu, k, kind=oracle.KIND_F32_MFMA)
    assert np.array_equal(i_gpu, i32)
    assert np.array_equal(s_gpu[valid], s32[valid].astype(np.float32)), np.abs(s_gpu[valid] - s32[valid]).max()


def test_scan_golden_fixture(gpu, oracle):
    """Committed golden vectors (tests/golden/search_N4096_D1024_Q32_seed1234.npz)."""
    import hashlib
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "search_N4096_D1024_Q32_seed1234.npz")
    g = np.load(path)
    xn = oracle.synthetic_unit_rows(int(g["n"]), int(g["dim"]), int(g["seed"]))
    assert hashlib.sha256(xn.tobytes()).digest() == g["xn_sha256"].tobytes(), "regenerated corpus differs"
    from rassengine_amd import ops
    corpus = gpu.from_numpy(xn).cuda()
    queries = gpu.from_numpy(g["q_raw"]).cuda()
    for k in (3, 5, 10):
        s, i = ops.scan_topk(corpus, queries, k)
        gpu.cuda.synchronize()
        assert np.array_equal(i.cpu().numpy(), g[f"ids_k{k}"])
        assert np.all(np.abs(s.cpu().numpy().astype(np.float64) - g[f"scores_k{k}"]) <= TOL_F64)


def test_scan_small_golden_fixture(gpu):
    """Stored rows incl. exact duplicates, tombstones and per-query patient filters (D=256)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "search_small_N512_D256.npz"))
    s, i = _run_scan(gpu, g["xn"], g["q_raw"], 10, 256, 256, tags=g["tags"], qfilter=g["qfilter"])
    assert np.array_equal(i, g["ids_k10"])
    valid = i >= 0
    assert np.all(np.abs(s[valid].astype(np.float64) - g["scores_k10"][valid]) <= TOL_F64)


def test_scan_ties_break_by_id(gpu, oracle):
    """Duplicated rows: equal scores must come back in ascending id order."""
    rng = np.random.default_rng(7)
    base = oracle.normalize_ref(rng.standard_normal((50, 1024), dtype=np.float32)).astype(np.float32)
    xn = np.concatenate([base, base[:10], base[:10], base[40:]], axis=0)  # rows 50..79 duplicate earlier rows
    q_raw = base[[3, 7, 45]] * 2.0
    s, i = _run_scan(gpu, xn, q_raw, 6, 1024, 1024)
    qn = oracle.normalize_c(q_raw)
    s64, i64 = oracle.search(xn, qn, 6, kind=oracle.KIND_F32_MFMA)
    assert np.array_equal(i, i64)
    # query 0 == row 3 == row 53 == row 63 : the three exact duplicates lead, ids ascending
    assert list(i[0, :3]) == [3, 53, 63]
    assert s[0, 0] == s[0, 1] == s[0, 2]


def test_scan_zero_rows_and_zero_query(gpu, oracle):
    rng = np.random.default_rng(11)
    x = rng.standard_normal((300, 1024), dtype=np.float32)
    x[5] = 0.0
    x[77] = 0.0
    xn = oracle.normalize_ref(x).astype(np.float32)
    q_raw = rng.standard_normal((3, 1024), dtype=np.float32)
    q_raw[1] = 0.0  # zero query: every score is 0, ranking is by id
    s, i = _run_scan(gpu, xn, q_raw, 10, 1024, 1024)
    assert list(i[1]) == list(range(10))
    assert np.all(s[1] == 0.0)
    qn = oracle.normalize_c(q_raw)
    s64, i64 = oracle.search(xn, qn, 10)
    assert np.array_equal(i[[0, 2]], i64[[0, 2]])


def test_scan_filter_and_tombstones(gpu, oracle):
    rng = np.random.default_rng(21)
    n = 5000
    xn = oracle.normalize_ref(rng.standard_normal((n, 1024), dtype=np.float32)).astype(np.float32)
    tags = rng.integers(0, 6, size=n).astype(np.int32)
    tags[rng.choice(n, 300, replace=False)] = -1  # tombstones
    q_raw = rng.standard_normal((9, 1024), dtype=np.float32)
    qfilter = np.array([-1, 0, 1, 2, 3, 4, 5, 99, -1], dtype=np.int32)  # 99 matches nothing
    s, i = _run_scan(gpu, xn, q_raw, 10, 1024, 1024, tags=tags, qfilter=qfilter, id_base=1_000_000)
    qn = oracle.normalize_c(q_raw)
    s64, i64 = oracle.search(xn, qn, 10, tags=tags, qfilter=qfilter, id_base=1_000_000)
    assert np.array_equal(i, i64)
    assert np.all(i[7] == -1) and np.all(np.isneginf(s[7]))
    live = i[i >= 0] - 1_000_000
    assert np.all(tags[live] != -1)


def test_scan_empty_corpus(gpu):
    from rassengine_amd import ops
    corpus = gpu.zeros((16, 1024), dtype=gpu.float32, device="cuda")
    q = gpu.randn((2, 1024), device="cuda")
    s, i = ops.scan_topk_packed(corpus, 0, q, 5)
    gpu.cuda.synchronize()
    assert np.all(i.cpu().numpy() == -1)
    assert np.all(np.isneginf(s.cpu().numpy()))


def test_scan_result_independent_of_sharding(gpu, oracle):
    """Row-sharding + merge == one scan, bit for bit (the multi-GPU invariant)."""
    from rassengine_amd import ops
    rng = np.random.default_rng(5)
    n = 9000
    xn = oracle.normalize_ref(rng.standard_normal((n, 1024), dtype=np.float32)).astype(np.float32)
    q = gpu.from_numpy(rng.standard_normal((16, 1024), dtype=np.float32)).cuda()
    corpus = gpu.from_numpy(xn).cuda()
    s_all, i_all = ops.scan_topk(corpus, q, 10)
    parts_s, parts_i = [], []
    bounds = [0, 1000, 1001, 4500, 9000]
    for a, b in zip(bounds[:-1], bounds[1:]):
        s, i = ops.scan_topk(corpus[a:b].contiguous(), q, 10, id_base=a)
        parts_s.append(s)
        parts_i.append(i)
    s_m, i_m = ops.topk_merge(gpu.stack(parts_s).contiguous(), gpu.stack(parts_i).contiguous())
    gpu.cuda.synchronize()
    assert gpu.equal(i_m, i_all)
    assert gpu.equal(s_m, s_all)


@pytest.mark.parametrize("n,dim,first", [(1, 1024, 0), (37, 1024, 5), (160, 384, 16), (50, 100, 3), (16, 7, 15)])
def test_pack_unpack_roundtrip(gpu, n, dim, first):
    """tile16 pack/unpack is a pure permutation: bit-exact round trip at any row offset, and
    the documented address formula holds."""
    import ctypes
    from rassengine_amd import ops, _native as N
    rng = np.random.default_rng(n + dim)
    x = rng.standard_normal((n, dim), dtype=np.float32)
    stride = (dim + 127) // 128 * 128
    total = (first + n + 15) // 16 * 16
    packed = gpu.full((total, stride), 7.0, dtype=gpu.float32, device="cuda")
    xd = gpu.from_numpy(x).cuda()
    N.check("pack", N.lib().rass_pack_rows_f32(ctypes.c_void_p(xd.data_ptr()), dim, ctypes.c_void_p(packed.data_ptr()),
                                               stride, first, n, dim, 0, None))
    back = ops.unpack_rows(packed, n, dim, first_row=first).cpu().numpy()
    assert np.array_equal(back, x)
    flat = packed.cpu().numpy().reshape(-1)
    for (r, c) in [(0, 0), (n - 1, dim - 1), (n // 2, dim // 3)]:
        rr = r + first
        off = (rr >> 4) * 16 * stride + (c >> 4) * 256 + ((((c >> 2) & 3) * 16 + (rr & 15)) * 4) + (c & 3)
        assert flat[off] == x[r, c]
    # rows of touched blocks outside [first, first+n) keep their previous content
    if first > 0:
        before = ops.unpack_rows(packed, first, dim, first_row=0).cpu().numpy()
        assert np.all(before == 7.0)


def test_gather_rows_equals_unpack(gpu):
    """rass_gather_rows_f32: scattered rows of a tile16 slab (k-means seeds) == the rows rass_unpack_rows_f32 returns; ids out
    of range leave their output row untouched."""
    import ctypes
    from rassengine_amd import _native as N_
    from rassengine_amd import ops
    torch = gpu
    rng = np.random.default_rng(9)
    for n, dim in ((1000, 1024), (77, 100), (300, 1536)):
        x = torch.from_numpy(rng.standard_normal((n, dim), dtype=np.float32)).cuda()
        packed = ops.pack_rows(x)
        ids = torch.from_numpy(np.concatenate([rng.integers(0, n, size=50), [0, n - 1, n, -1]]).astype(np.int64)).cuda()
        out = torch.full((ids.numel(), dim + 3), 7.0, device="cuda")
        N_.check("g", N_.lib().rass_gather_rows_f32(ctypes.c_void_p(packed.data_ptr()), packed.shape[1], n,
                                                    ctypes.c_void_p(ids.data_ptr()), ids.numel(), dim,
                                                    ctypes.c_void_p(out.data_ptr()), dim + 3,
                                                    ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
        torch.cuda.synchronize()
        assert torch.equal(out[:52, :dim], x[ids[:52]]) and bool((out[:, dim:] == 7.0).all()) and bool((out[52:] == 7.0).all())


def test_merge_matches_oracle(gpu, oracle):
    from rassengine_amd import ops
    rng = np.random.default_rng(9)
    for n_lists, nq, k in [(1, 1, 1), (2, 3, 5), (8, 16, 10), (256, 32, 32), (37, 5, 7)]:
        s = rng.standard_normal((n_lists, nq, k)).astype(np.float32)
        s = -np.sort(-s, axis=2)
        s[rng.random(s.shape) < 0.2] = np.float32(0.5)  # ties across lists
        s = -np.sort(-s, axis=2)
        ids = rng.permutation(n_lists * nq * k).reshape(n_lists, nq, k).astype(np.int64)
        empty = rng.random((n_lists, nq)) < 0.2
        ids[empty, k // 2:] = -1
        s[ids < 0] = -np.inf
        ms, mi = ops.topk_merge(gpu.from_numpy(s).cuda(), gpu.from_numpy(ids).cuda())
        gpu.cuda.synchronize()
        rs, ri = oracle.merge(s.astype(np.float64), ids)
        assert np.array_equal(mi.cpu().numpy(), ri)
        got = ms.cpu().numpy()
        assert np.array_equal(got[ri >= 0], rs[ri >= 0].astype(np.float32))


@pytest.mark.parametrize("n,dim,stride", [(1, 1024, 1024), (257, 1024, 1024), (100, 100, 128), (33, 7, 128),
                                          (1000, 384, 384), (5, 1023, 1024)])
def test_normalize_rows(gpu, oracle, n, dim, stride):
    """a4: within 2 ulp of the verbatim numpy expression; zero rows stay zero; padding is zero."""
    from rassengine_amd import ops
    rng = np.random.default_rng(dim + n)
    x = (rng.standard_normal((n, dim)) * rng.uniform(0.01, 100.0, size=(n, 1))).astype(np.float32)
    if n > 3:
        x[3] = 0.0
    ref = oracle.normalize_ref(x)
    out = ops.normalize_rows(gpu.from_numpy(x).cuda(), out_stride=stride).cpu().numpy()
    assert out.shape == (n, stride)
    assert np.all(out[:, dim:] == 0.0)
    np.testing.assert_allclose(out[:, :dim], ref, rtol=3e-7, atol=1e-30)
    if n > 3:
        assert np.all(out[3] == 0.0)


def test_strided_merge_of_packed_records(gpu, oracle):
    """The single-all-gather path: G packed (scores | ids) records merged in place."""
    import ctypes
    from rassengine_amd import _native as N
    from rassengine_amd.dist import HipShard
    rng = np.random.default_rng(12)
    G, nq, k = 5, 7, 10
    ids_off, size = HipShard.record_bytes(nq, k)
    assert ids_off % 8 == 0 and size % 8 == 0
    s = -np.sort(-rng.standard_normal((G, nq, k)).astype(np.float32), axis=2)
    ids = rng.permutation(G * nq * k).reshape(G, nq, k).astype(np.int64)
    buf = np.zeros((G, size), dtype=np.uint8)
    for g in range(G):
        buf[g, : nq * k * 4] = s[g].view(np.uint8).reshape(-1)
        buf[g, ids_off: ids_off + nq * k * 8] = ids[g].view(np.uint8).reshape(-1)
    d = gpu.from_numpy(buf.reshape(-1)).cuda()
    out_s = gpu.empty((nq, k), dtype=gpu.float32, device="cuda")
    out_i = gpu.empty((nq, k), dtype=gpu.int64, device="cuda")
    N.check("m", N.lib().rass_topk_merge_strided(ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(d.data_ptr() + ids_off),
                                                 size // 4, size // 8, G, nq, k, ctypes.c_void_p(out_s.data_ptr()),
                                                 ctypes.c_void_p(out_i.data_ptr()), None))
    gpu.cuda.synchronize()
    rs, ri = oracle.merge(s.astype(np.float64), ids)
    assert np.array_equal(out_i.cpu().numpy(), ri)
    assert np.array_equal(out_s.cpu().numpy(), rs.astype(np.float32))


def test_torch_library_ops_equal_the_plain_launchers(gpu, oracle):
    """torch.ops.rass.* (rassengine_amd/ops.py) run the same C-ABI launchers: same bits."""
    torch = gpu
    from rassengine_amd import ops
    rng = np.random.default_rng(3)
    xn = oracle.normalize_ref(rng.standard_normal((700, 384), dtype=np.float32)).astype(np.float32)
    q = torch.from_numpy(rng.standard_normal((9, 384), dtype=np.float32)).cuda()
    packed = ops.pack_rows(torch.from_numpy(xn).cuda())
    s0, i0 = ops.scan_topk_packed(packed, 700, q, 7)
    s1, i1 = torch.ops.rass.scan_topk_packed(packed, 700, q, 7)
    assert torch.equal(s0, s1) and torch.equal(i0, i1)
    assert torch.equal(torch.ops.rass.normalize_rows(q, 512), ops.normalize_rows(q, 512))
    ls = torch.stack([s0, s0 - 0.5]).contiguous()
    li = torch.stack([i0, i0 + 1000]).contiguous()
    a, b = torch.ops.rass.topk_merge(ls, li)
    c, d = ops.topk_merge(ls, li)
    assert torch.equal(a, c) and torch.equal(b, d) and torch.equal(b, i0)


def test_wide_index_end_to_end(gpu, oracle):
    """An index of 1 536-d rows (EMBED_DIM is an env knob of the reference, app/main.py:80; the encoder serves hidden sizes
    up to 2 048) through the product path: add (normalised on the GPU), tombstone, patient filter, 39 queries in one call
    (launch groups of 32 = two 16-query launches each), k = 70 in continuation passes, stored rows, save / load, the
    device batch API — ids equal the oracle's fp64 ranking, scores bit-equal to the emulation of the kernel's order."""
    import os
    import tempfile
    import torch
    from rassengine_amd.engine import Engine
    dim, n = 1536, 5000
    rng = np.random.default_rng(77)
    x = rng.standard_normal((n, dim), dtype=np.float32) * 2.0
    tags = rng.integers(1, 4, size=n).astype(np.int32)
    q = rng.standard_normal((39, dim), dtype=np.float32)
    eng = Engine(0, dim)
    try:
        idx = eng.open_index("wide")
        assert idx.row_stride == 1536 and idx.multi_tiles == 0
        assert idx.add(x[:3000], tags=tags[:3000]) == 0 and idx.add(x[3000:], tags=tags[3000:]) == 3000
        idx.delete(123)
        tags_live = tags.copy()
        tags_live[123] = -1
        xn = idx.get_rows(0, n)             # the rows as the GPU normalised and stored them (<= 1 ulp from numpy's)
        np.testing.assert_allclose(xn, oracle.normalize_ref(x), rtol=1e-6, atol=1e-9)
        qn = gpu.from_numpy(q).cuda()
        from rassengine_amd import ops
        qn = ops.normalize_rows(qn).cpu().numpy()
        qf = rng.integers(-1, 4, size=39).astype(np.int32)
        for k, filt in ((10, None), (10, qf), (70, qf)):
            s, i = idx.search(q, k, q_filter=filt)
            rs, ri = oracle.search(xn, qn, k, tags=tags_live, qfilter=filt, kind=oracle.KIND_F32_MFMA)
            assert np.array_equal(i, ri), (k, filt is not None)
            valid = ri >= 0
            assert np.array_equal(s[valid], rs[valid].astype(np.float32))
            r64, i64 = oracle.search(xn, qn, k, tags=tags_live, qfilter=filt, kind=oracle.KIND_F64)
            assert np.all(np.abs(s[valid].astype(np.float64) - r64[valid]) <= TOL_F64_WIDE)
        # device batch API (bench / sharded path): 64 queries in one call
        qd = torch.from_numpy(np.concatenate([q, q[:25]])).cuda()
        out_s = torch.empty((64, 10), dtype=torch.float32, device="cuda")
        out_i = torch.empty((64, 10), dtype=torch.int64, device="cuda")
        eng.set_stream(int(torch.cuda.current_stream().cuda_stream))
        idx.search_device_batch(qd.data_ptr(), 64, 10, out_s.data_ptr(), out_i.data_ptr())
        torch.cuda.synchronize()
        eng.reset_stream()
        s10, i10 = idx.search(q, 10)
        assert np.array_equal(out_i.cpu().numpy()[:39], i10) and np.array_equal(out_s.cpu().numpy()[:39], s10)
        assert np.array_equal(out_i.cpu().numpy()[39:], i10[:25])
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "wide.rass")
            idx.save(path)
            back = eng.load_index("wide-back", path)
            assert back.rows == n and back.count == n - 1
            s2, i2 = back.search(q, 10)
            assert np.array_equal(i2, i10) and np.array_equal(s2, s10)
    finally:
        eng.close()


def test_what_wide_rows_cannot_do_is_refused_loudly(gpu):
    """dim > 2 048 has no kernel; a wide-row index is served by the fp32 flat scan only: a bf16 corpus, the prefilter mode,
    the IVF build and cross-index batches are refused with an error, never answered wrongly."""
    from rassengine_amd._native import RassError
    from rassengine_amd.engine import Engine
    with pytest.raises((RassError, ValueError)):
        Engine(0, 2049)
    eng = Engine(0, 1280)
    try:
        idx = eng.open_index("w")
        idx.add(np.ones((40, 1280), dtype=np.float32))
        with pytest.raises(RassError):
            eng.open_index("wb", dtype="bf16")
        with pytest.raises(RassError):
            idx.set_prefilter(True)
        with pytest.raises(RassError):
            eng.search_multi([idx], np.ones((1, 1280), dtype=np.float32), 3)
        from rassengine_amd import ivf
        with pytest.raises((RassError, ValueError)):
            ivf.train_centroids(idx, nlist=4, iters=1)
    finally:
        eng.close()
