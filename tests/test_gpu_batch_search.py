"""rass_index_search_device_batch: many launch groups per engine call (one normalise launch, the groups' sample
passes, the scans, ONE grouped merge) must equal rass_index_search_device on consecutive groups of 32 queries bit
for bit — ragged last group, per-query filters, tombstones, caller-assigned ids, strided (packed-record) outputs,
bf16 corpora (group-by-group fallback) — and rass_topk_merge_strided_batch must equal the per-group strided merge."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _per_group(torch, ix, q, k, filt=None, id_base=0):
    n = q.shape[0]
    s = torch.empty((n, k), dtype=torch.float32, device="cuda")
    i = torch.empty((n, k), dtype=torch.int64, device="cuda")
    for g in range(0, n, 32):
        b = min(32, n - g)
        ix.search_device(q[g:g + b].data_ptr(), b, k, s[g:g + b].data_ptr(), i[g:g + b].data_ptr(), id_base=id_base,
                         d_q_filter_ptr=filt[g:g + b].data_ptr() if filt is not None else 0)
    torch.cuda.synchronize()
    return s.cpu().numpy(), i.cpu().numpy()


def _batch(torch, ix, q, k, filt=None, id_base=0):
    n = q.shape[0]
    s = torch.empty((n, k), dtype=torch.float32, device="cuda")
    i = torch.empty((n, k), dtype=torch.int64, device="cuda")
    ix.search_device_batch(q.data_ptr(), n, k, s.data_ptr(), i.data_ptr(), id_base=id_base,
                           d_q_filter_ptr=filt.data_ptr() if filt is not None else 0)
    torch.cuda.synchronize()
    return s.cpu().numpy(), i.cpu().numpy()


def _same(a, b):
    return np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))


@pytest.fixture(scope="module")
def small(gpu):
    from rassengine_amd.engine import Engine
    torch = gpu
    rng = np.random.default_rng(3)
    n, dim = 40_000, 1024
    x = rng.standard_normal((n, dim)).astype(np.float32)
    tags = rng.integers(1, 50, size=n).astype(np.int32)
    eng = Engine(0, dim)
    eng.set_stream(int(torch.cuda.current_stream().cuda_stream))
    ix = eng.open_index("batch")
    ix.add(x, tags=tags)
    for r in (0, 17, 39_999, 20_000):
        ix.delete(r)
    yield torch, eng, ix
    eng.close()


@pytest.mark.parametrize("nq,k", [(33, 10), (64, 10), (100, 7), (1024, 10), (70, 32), (17, 5), (32, 10)])
def test_batch_equals_groups(small, nq, k):
    torch, eng, ix = small
    g = torch.Generator(device="cuda"); g.manual_seed(nq * 31 + k)
    q = torch.randn((nq, 1024), generator=g, device="cuda")
    for mode in ("0", "force"):      # without / with the sample floor (40 000 rows >= 2 samples when forced)
        os.environ["RASS_SCAN_SAMPLE_FLOOR"] = mode
        try:
            assert _same(_batch(torch, ix, q, k), _per_group(torch, ix, q, k)), mode
        finally:
            os.environ.pop("RASS_SCAN_SAMPLE_FLOOR", None)


def test_batch_with_filters_and_id_base(small):
    torch, eng, ix = small
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    q = torch.randn((90, 1024), generator=g, device="cuda")
    filt = torch.randint(1, 50, (90,), dtype=torch.int32, device="cuda")
    filt[3] = -1
    filt[50] = 9999     # matches nothing
    a = _batch(torch, ix, q, 10, filt=filt, id_base=7_000_000)
    b = _per_group(torch, ix, q, 10, filt=filt, id_base=7_000_000)
    assert _same(a, b)
    assert np.all(a[1][50] == -1) and a[1][3, 0] >= 7_000_000


def test_batch_strided_outputs_are_packed_records(small):
    """The N > 1 bench path: group g's scores / ids land inside its packed record."""
    torch, eng, ix = small
    from rassengine_amd.dist import HipShard
    k, groups = 10, 4
    g = torch.Generator(device="cuda"); g.manual_seed(6)
    q = torch.randn((32 * groups, 1024), generator=g, device="cuda")
    shard = HipShard(ix, id_base=500)
    ids_off, size = HipShard.record_bytes(32, k)
    recs = torch.zeros((groups * size,), dtype=torch.uint8, device="cuda")
    shard.search_local_packed_batch_into(q, k, 32, recs)
    ref = torch.zeros((groups * size,), dtype=torch.uint8, device="cuda")
    for j in range(groups):
        shard.search_local_packed_into(q[j * 32:(j + 1) * 32], k, ref[j * size:(j + 1) * size])
    torch.cuda.synchronize()
    assert torch.equal(recs, ref)
    # a two-"rank" gathered buffer (the same shard twice): grouped merge == per-group merges
    gathered = torch.cat([recs, recs])
    out_s = torch.empty((32 * groups, k), dtype=torch.float32, device="cuda")
    out_i = torch.empty((32 * groups, k), dtype=torch.int64, device="cuda")
    shard.merge_packed_batch(gathered, 2, groups, 32, k, out_s, out_i)
    ref_s = torch.empty_like(out_s)
    ref_i = torch.empty_like(out_i)
    for j in range(groups):
        shard.merge_packed_group(gathered, 2, j, groups, 32, k, ref_s[j * 32:(j + 1) * 32], ref_i[j * 32:(j + 1) * 32])
    torch.cuda.synchronize()
    assert torch.equal(out_i, ref_i) and torch.equal(out_s, ref_s)
    # duplicates of one list: every id appears twice, best first
    assert torch.equal(out_i[:, 0], out_i[:, 1])


def test_batch_on_bf16_corpus_falls_back_group_by_group(gpu):
    from rassengine_amd.engine import Engine
    torch = gpu
    eng = Engine(0, 1024)
    try:
        eng.set_stream(int(torch.cuda.current_stream().cuda_stream))
        ix = eng.open_index("b16", capacity_rows=20_000, dtype="bf16")
        ix.fill_synthetic(20_000, seed=4)
        q = torch.randn((70, 1024), device="cuda")
        assert _same(_batch(torch, ix, q, 10), _per_group(torch, ix, q, 10))
    finally:
        eng.close()


def test_batch_full_size_equals_groups(gpu):
    """1M rows, 1 024 queries (the bench step), sample floor at its default."""
    from rassengine_amd.engine import Engine
    torch = gpu
    eng = Engine(0, 1024)
    try:
        eng.set_stream(int(torch.cuda.current_stream().cuda_stream))
        ix = eng.open_index("full", capacity_rows=1_000_000)
        ix.fill_synthetic(1_000_000, seed=11)
        q = torch.randn((1024, 1024), device="cuda")
        assert _same(_batch(torch, ix, q, 10), _per_group(torch, ix, q, 10))
    finally:
        eng.close()


def test_batch_argument_errors(small):
    torch, eng, ix = small
    from rassengine_amd import _native as N
    q = torch.randn((40, 1024), device="cuda")
    s = torch.empty((40, 10), dtype=torch.float32, device="cuda")
    i = torch.empty((40, 10), dtype=torch.int64, device="cuda")
    with pytest.raises(Exception):
        ix.search_device_batch(q.data_ptr(), 0, 10, s.data_ptr(), i.data_ptr())
    with pytest.raises(Exception):
        ix.search_device_batch(q.data_ptr(), 40, 33, s.data_ptr(), i.data_ptr())
    with pytest.raises(Exception):
        ix.search_device_batch(q.data_ptr(), 40, 10, s.data_ptr(), i.data_ptr(), out_scores_group_stride=100)
    ix.search_device_batch(q.data_ptr(), 40, 10, s.data_ptr(), i.data_ptr())   # still works afterwards
    torch.cuda.synchronize()
