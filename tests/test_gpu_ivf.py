"""GPU tests of the IVF index (K9): probing every list reproduces the exact flat result,
recall@10 grows with nprobe on clustered data, filters / tombstones carry over, and the
fine scan touches only the probed lists."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _clustered(rng, n, dim, centres, sigma):
    c = rng.standard_normal((centres, dim)).astype(np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    lab = rng.integers(0, centres, size=n)
    x = c[lab] + sigma * rng.standard_normal((n, dim)).astype(np.float32) / np.sqrt(dim)
    return x.astype(np.float32), c, lab


@pytest.fixture(scope="module")
def built(gpu):
    from rassengine_amd.engine import Engine
    from rassengine_amd.ivf import IvfIndex
    rng = np.random.default_rng(42)
    n, dim = 30000, 1024
    x, centres, _ = _clustered(rng, n, dim, 200, 1.0)
    tags = rng.integers(1, 5, size=n).astype(np.int32)
    eng = Engine(0, dim)
    flat = eng.open_index("ivf-src")
    flat.add(x, tags=tags)
    for r in (5, 77, 12345):
        flat.delete(r)
    ivf = IvfIndex.build(flat, nlist=128, iters=8, seed=3)
    q = centres[rng.integers(0, 200, size=50)] + 0.8 * rng.standard_normal((50, dim)).astype(np.float32) / np.sqrt(dim)
    yield eng, flat, ivf, q.astype(np.float32), tags
    ivf.close()
    eng.close()


def test_probe_all_lists_equals_flat(built):
    eng, flat, ivf, q, _ = built
    assert ivf.nlist == 128 and ivf.rows == flat.count
    # nlist = 128 > 32 lists per probe: cover all lists in 4 disjoint probes?  Not possible through
    # the API (probe picks the best lists), so compare at the largest nprobe against flat recall
    # and, for exactness, on a second index with nlist <= 32.
    from rassengine_amd.ivf import IvfIndex
    ivf32 = IvfIndex.build(flat, nlist=32, iters=5, seed=1)
    try:
        s_f, i_f = flat.search(q, 10)
        s_i, i_i, scanned = ivf32.search(q, 10, nprobe=32)
        assert np.array_equal(i_i, i_f)
        assert np.array_equal(s_i, s_f)       # same kernel, same fmaf order: bit-identical scores
        assert scanned == 2 * flat.count      # 50 queries = 2 batches, each touching every live row once
    finally:
        ivf32.close()


def test_recall_grows_with_nprobe(built):
    eng, flat, ivf, q, _ = built
    s_f, i_f = flat.search(q, 10)
    recalls, scanned = [], []
    for nprobe in (1, 2, 4, 8, 16, 32):
        s, i, sc = ivf.search(q, 10, nprobe=nprobe)
        recalls.append(np.mean([len(set(i[r]) & set(i_f[r])) / 10 for r in range(q.shape[0])]))
        scanned.append(sc)
        found = i >= 0
        # every returned score is the true cosine of that row: compare with the flat scores of the same ids
        for r in range(q.shape[0]):
            common = {int(a): float(b) for a, b in zip(i_f[r], s_f[r])}
            for a, b in zip(i[r][found[r]], s[r][found[r]]):
                if int(a) in common:
                    assert b == common[int(a)]
    assert all(b >= a - 1e-9 for a, b in zip(recalls, recalls[1:])), recalls
    assert recalls[-1] >= 0.95, recalls
    assert recalls[0] >= 0.3, recalls
    assert all(b >= a for a, b in zip(scanned, scanned[1:]))
    # one list per query: a batch of b queries touches at most b lists, so the rows scanned by the two batches
    # (32 + 18 queries) cannot exceed the 32 + 18 longest lists.  (Round 1 had `< 0.2 * 2 * count` here, which
    # failed — queries fall into LONG lists more often than into short ones, size-biased sampling — and was
    # loosened to 0.3; the exact equality `scanned == rows of the union of the probed lists` is pinned in
    # tests/test_gpu_cfg5.py against the oracle's coarse ranking.)
    longest = np.sort(ivf.list_sizes)[::-1]
    assert scanned[0] <= int(longest[:32].sum() + longest[:18].sum())
    assert scanned[0] >= int(np.sort(ivf.list_sizes)[:1].sum())


def test_wide_probe_threshold_path(built):
    """nprobe > 32 goes through the full centroid-score matrix + radix-select threshold: probing
    all 128 lists must reproduce the flat result exactly; 64 lists must not be worse than 32."""
    eng, flat, ivf, q, _ = built
    s_f, i_f = flat.search(q, 10)
    s_all, i_all, scanned = ivf.search(q, 10, nprobe=128)
    assert np.array_equal(i_all, i_f) and np.array_equal(s_all, s_f)
    assert scanned == 2 * flat.count
    s_big, i_big, _ = ivf.search(q, 10, nprobe=100000)      # capped at nlist
    assert np.array_equal(i_big, i_f)
    rec = {}
    for nprobe in (32, 33, 64, 96):
        s, i, sc = ivf.search(q, 10, nprobe=nprobe)
        rec[nprobe] = (np.mean([len(set(i[r]) & set(i_f[r])) / 10 for r in range(q.shape[0])]), sc)
    assert rec[33][0] >= rec[32][0] - 1e-9 and rec[64][0] >= rec[33][0] - 1e-9 and rec[96][0] >= rec[64][0] - 1e-9
    assert rec[32][1] <= rec[33][1] <= rec[64][1] <= rec[96][1] <= 2 * flat.count


def test_ivf_filters_and_tombstones(built):
    eng, flat, ivf, q, tags = built
    qf = np.array([(r % 4) + 1 for r in range(q.shape[0])], dtype=np.int32)
    s, i, _ = ivf.search(q, 10, nprobe=32, q_filter=qf)
    for r in range(q.shape[0]):
        live = i[r][i[r] >= 0]
        assert np.all(tags[live] == qf[r])
        assert not set(live.tolist()) & {5, 77, 12345}
    s_f, i_f = flat.search(q, 10, q_filter=qf)
    rec = np.mean([len(set(i[r]) & set(i_f[r])) / 10 for r in range(q.shape[0])])
    assert rec >= 0.9


def test_ivf_empty_lists_and_small_index(gpu):
    """More lists than distinct points: empty lists must be harmless."""
    from rassengine_amd.engine import Engine
    from rassengine_amd.ivf import IvfIndex
    import torch
    rng = np.random.default_rng(1)
    base = rng.standard_normal((8, 1024)).astype(np.float32)
    x = np.repeat(base, 20, axis=0) + 1e-3 * rng.standard_normal((160, 1024)).astype(np.float32)
    eng = Engine(0, 1024)
    try:
        flat = eng.open_index("ivf-small")
        flat.add(x)
        cent = torch.from_numpy(np.concatenate([base, rng.standard_normal((24, 1024)).astype(np.float32)])).cuda()
        cent = cent / cent.norm(dim=1, keepdim=True)
        ivf = IvfIndex.build(flat, nlist=32, centroids=cent)
        assert int((ivf.list_sizes == 0).sum()) >= 20
        s, i, scanned = ivf.search(base, 5, nprobe=3)
        s_f, i_f = flat.search(base, 5)
        assert np.array_equal(i, i_f)
        ivf.close()
    finally:
        eng.close()


def test_ivf_save_load_roundtrip(built, tmp_path):
    """rass_ivf_save / rass_ivf_load: the whole device state travels (no flat index, no re-training needed);
    the loaded shard answers bit for bit like the one that was saved; corrupt files are refused."""
    from rassengine_amd.ivf import IvfIndex
    from rassengine_amd._native import RassError
    eng, flat, ivf, q, tags = built
    path = str(tmp_path / "shard0.ivf")
    ivf.save(path)
    import os
    assert os.path.exists(path) and not os.path.exists(path + ".tmp")
    back = IvfIndex.load(eng, path)
    try:
        assert back.nlist == ivf.nlist and back.rows == ivf.rows
        qf = np.array([(r % 4) + 1 for r in range(q.shape[0])], dtype=np.int32)
        for nprobe, f in ((1, None), (8, None), (128, None), (32, qf)):
            s0, i0, sc0 = ivf.search(q, 10, nprobe, q_filter=f)
            s1, i1, sc1 = back.search(q, 10, nprobe, q_filter=f)
            assert np.array_equal(i0, i1) and np.array_equal(s0, s1) and sc0 == sc1
    finally:
        back.close()
    raw = open(path, "rb").read()
    bad = str(tmp_path / "truncated.ivf")
    open(bad, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(RassError):
        IvfIndex.load(eng, bad)
    bad2 = str(tmp_path / "garbage.ivf")
    open(bad2, "wb").write(b"not an ivf file" * 100)
    with pytest.raises(RassError):
        IvfIndex.load(eng, bad2)
    # a list table that points outside the slab is refused too (it would send the probe out of bounds)
    import struct
    hdr = 8 + 4 * 4 + 8 * 5 + 8            # header + (file versions 3 / 4) the covered source rows
    tampered = bytearray(raw)
    tampered[hdr:hdr + 4] = struct.pack("<i", 5)            # list 0 no longer starts at tile 0
    bad3 = str(tmp_path / "tampered.ivf")
    open(bad3, "wb").write(bytes(tampered))
    with pytest.raises(RassError):
        IvfIndex.load(eng, bad3)


@pytest.mark.parametrize("slab", ["f32", "bf16"])
def test_batch_of_launch_groups_equals_group_by_group(built, slab):
    """rass_ivf_search_device_batch (one grouped coarse scan over the centroid slab, one plan launch with the coarse lists
    merged inside, G fine scans, one grouped merge) == rass_ivf_search_device on consecutive groups of 32 queries: same
    lists probed, ids and scores BIT FOR BIT, same scanned rows per group — 1 .. 1 024 queries, nprobe 1 .. 32 (and the deep
    threshold path group by group), per-query filters."""
    import torch
    from rassengine_amd.ivf import IvfIndex
    eng, flat, ivf0, q, tags = built
    ivf = ivf0 if slab == "f32" else IvfIndex.build(flat, nlist=128, centroids=ivf0.centroids, dtype="bf16")
    try:
        dev = torch.device("cuda", 0)
        g = torch.Generator(device=dev)
        g.manual_seed(12)
        qd = torch.cat([torch.from_numpy(q).to(dev), torch.randn((1024 - q.shape[0], 1024), generator=g, device=dev)]).contiguous()
        filt = torch.tensor([(r % 5) if r % 3 else -1 for r in range(1024)], dtype=torch.int32, device=dev)
        for nq, k, nprobe, use_f in ((32, 10, 1, False), (100, 10, 2, False), (1024, 10, 2, False), (77, 5, 8, True),
                                     (1, 32, 32, False), (200, 10, 32, True), (64, 10, 64, False)):
            out_s = torch.empty((nq, k), device=dev)
            out_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
            groups = (nq + 31) // 32
            sc = torch.zeros((groups,), dtype=torch.int64, device=dev)
            ivf.search_device_batch(qd.data_ptr(), nq, k, nprobe, out_s.data_ptr(), out_i.data_ptr(),
                                    filt.data_ptr() if use_f else 0, sc.data_ptr())
            eng.synchronize()
            ref_s = torch.empty((nq, k), device=dev)
            ref_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
            for b0 in range(0, nq, 32):
                b = min(32, nq - b0)
                ivf.search_device(qd[b0:b0 + b].data_ptr(), b, k, nprobe, ref_s[b0:b0 + b].data_ptr(), ref_i[b0:b0 + b].data_ptr(),
                                  filt[b0:b0 + b].data_ptr() if use_f else 0)
            eng.synchronize()
            assert torch.equal(out_i, ref_i) and torch.equal(out_s, ref_s), (slab, nq, k, nprobe, use_f)
            # scanned rows per group == what the host API reports for that group
            qh = qd[:min(nq, 64)].cpu().numpy()
            fh = filt[:min(nq, 64)].cpu().numpy() if use_f else None
            for gidx in range(min(groups, 2)):
                lo, hi = 32 * gidx, min(nq, 32 * gidx + 32)
                _, _, want = ivf.search(qh[lo:hi], k, nprobe, q_filter=None if fh is None else fh[lo:hi])
                assert int(sc[gidx]) == want, (slab, nq, nprobe, gidx)
    finally:
        if ivf is not ivf0:
            ivf.close()
