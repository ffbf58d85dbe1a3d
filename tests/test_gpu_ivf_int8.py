"""GPU tests of the IVF over an int8 slab (rass_ivf_build_ex(..., RASS_I8); SURVEY §8f-4 "bf16 (or int8)" for cfg 5).

The fine scan reads the int8 copy of the list-ordered rows (a quarter of the bytes) for 32 candidates per query; those are
rescored EXACTLY from the IVF's fp32 copy.  What is pinned:
 * every returned (id, score) pair is the fp32 IVF's pair for that row, bit for bit (the re-rank runs the flat kernel's
   fmaf order), and on these corpora the lists are IDENTICAL to the fp32 IVF's at every nprobe (the probed lists' true
   top-k lies inside the int8 top-32) — hence nprobe = nlist reproduces the flat fp32 index;
 * `scanned` (rows of the union of a batch's probed lists) equals the fp32 IVF's over the same centroids;
 * plain and masked filters, tombstones before and after the build, save / load (the int8 copy is rebuilt), the batch call
   ≡ group by group, the delta path behind ``IvfBackedIndex`` (k > 16 takes the exact scan), what is refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
    rng = np.random.default_rng(777)
    n, dim = 30000, 1024
    x, centres = _clustered(rng, n, dim, 200, 1.0)
    tags = (rng.integers(1, 5, size=n) | (rng.integers(0, 2, size=n) << 24)).astype(np.int32)   # patient code | doc_type bit
    eng = Engine(0, dim)
    flat = eng.open_index("ivf8-src")
    flat.add(x, tags=tags)
    for r in DEAD:
        flat.delete(r)
    cent = train_centroids(flat, 128, train_rows=0, iters=8, seed=3)
    ivf_8 = IvfIndex.build(flat, nlist=128, centroids=cent, dtype="int8")
    ivf_f = IvfIndex.build(flat, nlist=128, centroids=cent)
    q = centres[rng.integers(0, 200, size=50)] + 0.8 * rng.standard_normal((50, dim)).astype(np.float32) / np.sqrt(dim)
    yield eng, flat, ivf_8, ivf_f, q.astype(np.float32), tags
    ivf_8.close()
    ivf_f.close()
    eng.close()


def _same(a, b):
    return np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))


def test_every_list_probed_equals_the_flat_fp32_index(built):
    eng, flat, ivf_8, ivf_f, q, tags = built
    from rassengine_amd.ivf import IvfIndex
    assert ivf_8.dtype == "int8" and ivf_8.max_k == 16 and ivf_f.max_k == 32
    assert ivf_8.nlist == 128 and ivf_8.rows == flat.count
    s_f, i_f = flat.search(q, 10)
    s_all, i_all, scanned = ivf_8.search(q, 10, nprobe=128)          # threshold path (nprobe > 32), every list
    assert _same((s_all, i_all), (s_f, i_f))
    assert scanned == 2 * flat.count
    ivf32 = IvfIndex.build(flat, nlist=32, iters=5, seed=1, dtype="int8")
    try:
        for nq, k in ((50, 10), (1, 16), (17, 7)):                   # top-k path; the 16-query kernel variant, a ragged half
            s_i, i_i, _ = ivf32.search(q[:nq], k, nprobe=32)
            assert _same((s_i, i_i), flat.search(q[:nq], k)), (nq, k)
    finally:
        ivf32.close()


@pytest.mark.parametrize("nprobe", [1, 4, 32, 33, 64])
def test_partial_probe_equals_the_fp32_ivf(built, nprobe):
    eng, flat, ivf_8, ivf_f, q, tags = built
    assert np.array_equal(ivf_8.assign, ivf_f.assign)
    s8, i8, sc8 = ivf_8.search(q, 10, nprobe=nprobe)
    sf, i_f, scf = ivf_f.search(q, 10, nprobe=nprobe)
    assert sc8 == scf
    assert _same((s8, i8), (sf, i_f)), nprobe


def test_filters_masks_and_tombstones(built):
    eng, flat, ivf_8, ivf_f, q, tags = built
    qf = np.array([(r % 4) + 1 if r % 3 else -1 for r in range(50)], dtype=np.int32)
    assert _same(ivf_8.search(q, 8, nprobe=16, q_filter=qf)[:2], ivf_f.search(q, 8, nprobe=16, q_filter=qf)[:2])
    # masked compare (patient code AND / OR the doc_type bit) through the delta entry point (no delta: covered == rows)
    m = np.array([0x00ffffff if r % 2 else 0x01000000 for r in range(50)], dtype=np.int32)
    f = np.array([(r % 4) + 1 if r % 2 else 0x01000000 for r in range(50)], dtype=np.int32)
    a = ivf_8.search_delta(flat, q, 8, 16, f, m)
    b = ivf_f.search_delta(flat, q, 8, 16, f, m)
    assert _same(a[:2], b[:2])
    for r in range(50):
        got = a[1][r][a[1][r] >= 0]
        assert np.all((tags[got] & m[r]) == f[r])
        assert not set(got) & set(DEAD)
    # a tombstone AFTER the build reaches both copies' tag array
    victim = int(a[1][0][0])
    try:
        flat.delete(victim)
        ivf_8.delete(victim)
        ivf_f.delete(victim)
        a2 = ivf_8.search_delta(flat, q, 8, 16, f, m)
        assert victim not in a2[1] and _same(a2[:2], ivf_f.search_delta(flat, q, 8, 16, f, m)[:2])
    finally:
        pass


def test_save_load_rebuilds_the_int8_copy(built, tmp_path):
    eng, flat, ivf_8, ivf_f, q, tags = built
    from rassengine_amd.ivf import IvfIndex
    p = str(tmp_path / "i8.ivf")
    ivf_8.save(p)
    back = IvfIndex.load(eng, p)
    try:
        assert back.dtype == "int8" and back.rows == ivf_8.rows and back.covered_rows == ivf_8.covered_rows
        for nprobe in (2, 128):
            assert _same(back.search(q, 10, nprobe)[:2], ivf_8.search(q, 10, nprobe)[:2])
    finally:
        back.close()


def test_batch_call_equals_group_by_group(built):
    import torch
    eng, flat, ivf_8, ivf_f, q, tags = built
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    qd = torch.cat([torch.from_numpy(q).to(dev), torch.randn((1024 - q.shape[0], 1024), generator=g, device=dev)]).contiguous()
    filt = torch.tensor([(r % 5) if r % 3 else -1 for r in range(1024)], dtype=torch.int32, device=dev)
    for nq, k, nprobe, use_f in ((32, 10, 1, False), (100, 10, 2, False), (1024, 10, 2, False), (77, 5, 8, True),
                                 (1, 16, 32, False), (200, 10, 32, True), (64, 10, 64, False)):
        out_s = torch.empty((nq, k), device=dev)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
        sc = torch.zeros(((nq + 31) // 32,), dtype=torch.int64, device=dev)
        ivf_8.search_device_batch(qd.data_ptr(), nq, k, nprobe, out_s.data_ptr(), out_i.data_ptr(), filt.data_ptr() if use_f else 0,
                                  sc.data_ptr())
        eng.synchronize()
        ref_s = torch.empty((nq, k), device=dev)
        ref_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
        f_s = torch.empty((nq, k), device=dev)
        f_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
        for b0 in range(0, nq, 32):
            b = min(32, nq - b0)
            ivf_8.search_device(qd[b0:b0 + b].data_ptr(), b, k, nprobe, ref_s[b0:b0 + b].data_ptr(), ref_i[b0:b0 + b].data_ptr(),
                                filt[b0:b0 + b].data_ptr() if use_f else 0)
            ivf_f.search_device(qd[b0:b0 + b].data_ptr(), b, k, nprobe, f_s[b0:b0 + b].data_ptr(), f_i[b0:b0 + b].data_ptr(),
                                filt[b0:b0 + b].data_ptr() if use_f else 0)
        eng.synchronize()
        assert torch.equal(out_i, ref_i) and torch.equal(out_s, ref_s), (nq, k, nprobe, use_f)
        # against the fp32 IVF: the random (unclustered) queries rank rows of whatever lists they probe — every returned
        # pair must be the fp32 IVF's pair, and nearly all lists identical
        same_rows = (out_i == f_i)
        assert torch.equal(out_s[same_rows], f_s[same_rows])
        assert same_rows.float().mean().item() >= 0.98, (nq, nprobe, same_rows.float().mean().item())


def test_behind_the_boundary_with_a_delta(gpu):
    """IvfBackedIndex with an int8 slab: appends after the build (flat delta, scanned exactly), overwrite-style tombstones on
    both sides; every list probed + delta ≡ the exact flat scan for k <= 16; k > 16 takes the exact scan by itself."""
    from rassengine_amd.engine import Engine
    from rassengine_amd.ivf import IvfBackedIndex, IvfPolicy
    rng = np.random.default_rng(31)
    x, centres = _clustered(rng, 24_000, 1024, 150, 1.0)
    eng = Engine(0, 1024)
    try:
        ix = IvfBackedIndex(eng.open_index("ivf8-backed"), IvfPolicy.manual(nprobe=64))
        ix.add(x[:20_000])
        ix.build_ivf(nlist=64, dtype="int8")
        assert ix.ivf.dtype == "int8" and ix.covered == 20_000
        ix.add(x[20_000:])                       # the delta
        for r in (3, 19_999, 20_001, 23_999):
            ix.delete(r)
        q = (centres[:40] + 0.8 * rng.standard_normal((40, 1024)).astype(np.float32) / 32).astype(np.float32)
        for k in (1, 10, 16, 20, 40):
            a = ix.search(q, k)
            b = ix.search(q, k, exact=True)
            assert _same(a, b), k
        s2, i2 = ix.search(q, 10, nprobe=2)      # a partial probe still sees the whole delta
        sf, i_f = ix.search(q, 10, exact=True)
        hit = np.mean([len(set(i2[r]) & set(i_f[r])) / 10 for r in range(40)])
        assert hit >= 0.9, hit
    finally:
        eng.close()


def test_what_an_int8_slab_does_not_serve(built):
    from rassengine_amd._native import RassError
    eng, flat, ivf_8, ivf_f, q, tags = built
    with pytest.raises(RassError):
        ivf_8.search(q, 17, nprobe=4)            # 32 candidates per query: k <= 16
    with pytest.raises(ValueError):
        from rassengine_amd.ivf import IvfIndex
        IvfIndex.build(flat, nlist=16, dtype="fp8")
