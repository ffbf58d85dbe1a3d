"""GPU tests of the engine / index objects behind the C ABI (rass_engine_* / rass_index_*):
the write path (normalise + pack on add, growth, tombstones), the host search API and
persistence, all checked against the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def engine(gpu):
    from rassengine_amd.engine import Engine
    eng = Engine(device=0, dim=1024)
    yield eng
    eng.close()


def test_add_in_odd_batches_and_search(engine, oracle):
    rng = np.random.default_rng(3)
    idx = engine.open_index("t-odd")  # no initial capacity: exercises growth
    chunks = [1, 15, 16, 17, 1000, 3, 2500]
    xs, tags = [], []
    first_expected = 0
    for c in chunks:
        x = rng.standard_normal((c, 1024), dtype=np.float32) * rng.uniform(0.1, 20.0)
        t = rng.integers(0, 4, size=c).astype(np.int32)
        first = idx.add(x, tags=t)
        assert first == first_expected
        first_expected += c
        xs.append(x)
        tags.append(t)
    x = np.concatenate(xs)
    tags = np.concatenate(tags)
    n = x.shape[0]
    assert idx.rows == n and idx.count == n
    xn = oracle.normalize_ref(x).astype(np.float32)

    # stored rows are the normalised rows (within 2 ulp of the numpy expression)
    for r in (0, 1, 16, 17, 1031, n - 1):
        np.testing.assert_allclose(idx.get_row(r), xn[r], rtol=3e-7, atol=1e-30)

    q = rng.standard_normal((70, 1024), dtype=np.float32)  # > 32: host API batches
    qn = oracle.normalize_ref(q).astype(np.float32)
    s, i = idx.search(q, 10)
    rs, ri = oracle.search(xn, qn, 10)
    assert np.array_equal(i, ri)
    assert np.all(np.abs(s - rs) <= 2e-6)

    qf = rng.integers(-1, 4, size=70).astype(np.int32)
    s, i = idx.search(q, 5, q_filter=qf)
    rs, ri = oracle.search(xn, qn, 5, tags=tags, qfilter=qf)
    assert np.array_equal(i, ri)


def test_delete_is_overwrite_semantics(engine, oracle):
    """Re-adding a doc_id overwrites in the reference (_id=doc_id, app/main.py:1260):
    tombstone the old row, append the new one."""
    rng = np.random.default_rng(4)
    idx = engine.open_index("t-del")
    x = rng.standard_normal((200, 1024), dtype=np.float32)
    idx.add(x)
    q = x[[10, 20]] + 0.01 * rng.standard_normal((2, 1024), dtype=np.float32)
    s, i = idx.search(q, 3)
    assert i[0, 0] == 10 and i[1, 0] == 20
    idx.delete(10)
    idx.delete(10)  # idempotent
    assert idx.count == 199 and idx.rows == 200
    new_row = idx.add(x[10:11] * 2.0)  # same direction, re-added
    assert new_row == 200
    s, i = idx.search(q, 3)
    assert i[0, 0] == 200 and 10 not in i[0]
    tags = np.zeros(201, dtype=np.int32)
    tags[10] = -1
    xn = oracle.normalize_ref(np.concatenate([x, x[10:11] * 2.0])).astype(np.float32)
    rs, ri = oracle.search(xn, oracle.normalize_ref(q).astype(np.float32), 3, tags=tags)
    assert np.array_equal(i, ri)
    with pytest.raises(Exception):
        idx.delete(5000)


def test_empty_index_and_has_any_data(engine):
    idx = engine.open_index("t-empty")
    assert idx.count == 0
    s, i = idx.search(np.ones((2, 1024), dtype=np.float32), 4)
    assert np.all(i == -1) and np.all(np.isneginf(s))
    # open is a lookup when the index exists
    assert engine.open_index("t-empty") is idx


def test_save_load_roundtrip(engine, oracle, tmp_path):
    rng = np.random.default_rng(6)
    idx = engine.open_index("t-save")
    x = rng.standard_normal((777, 1024), dtype=np.float32)
    tags = rng.integers(0, 3, size=777).astype(np.int32)
    idx.add(x, tags=tags)
    idx.delete(5)
    path = os.path.join(str(tmp_path), "shard.rass")
    idx.save(path)
    idx2 = engine.load_index("t-save-copy", path)
    assert idx2.rows == 777 and idx2.count == 776
    q = rng.standard_normal((8, 1024), dtype=np.float32)
    qf = np.array([-1, 0, 1, 2, -1, 0, 1, 2], dtype=np.int32)
    s1, i1 = idx.search(q, 10, q_filter=qf)
    s2, i2 = idx2.search(q, 10, q_filter=qf)
    assert np.array_equal(i1, i2) and np.array_equal(s1, s2)
    for r in (0, 5, 776):
        assert np.array_equal(idx.get_row(r), idx2.get_row(r))
    # a file shorter than its header claims, or with an absurd row count, is refused before any
    # allocation is sized from it
    from rassengine_amd._native import RassError
    blob = open(path, "rb").read()
    cut = os.path.join(str(tmp_path), "cut.rass")
    open(cut, "wb").write(blob[: len(blob) // 2])
    with pytest.raises(RassError, match="truncated"):
        engine.load_index("t-save-cut", cut)
    huge = bytearray(blob)
    huge[24:32] = (1 << 45).to_bytes(8, "little")          # SaveHeader.rows
    bad = os.path.join(str(tmp_path), "huge.rass")
    open(bad, "wb").write(bytes(huge))
    with pytest.raises(RassError, match="truncated"):
        engine.load_index("t-save-huge", bad)
    with pytest.raises(RassError, match="not a rass index"):
        open(bad, "wb").write(b"garbage" * 10)
        engine.load_index("t-save-garbage", bad)


def test_errors_are_reported_not_thrown(engine):
    from rassengine_amd._native import RassError
    idx = engine.open_index("t-err")
    idx.add(np.ones((4, 1024), dtype=np.float32))
    with pytest.raises(RassError) as e:
        idx.search(np.ones((1, 1024), dtype=np.float32), 4097)  # k > RASS_MAX_K_MULTIPASS
    assert "k must be" in str(e.value)
    with pytest.raises(RassError):
        idx.search(np.ones((1, 1024), dtype=np.float32), 0)
    s33, i33 = idx.search(np.ones((1, 1024), dtype=np.float32), 33)   # k > RASS_MAX_K is served in passes
    assert sorted(i33[0][:4].tolist()) == [0, 1, 2, 3] and np.all(i33[0][4:] == -1)
    with pytest.raises(RassError):
        idx.add(np.ones((1, 1024), dtype=np.float32), tags=np.array([-3], dtype=np.int32))
    with pytest.raises(ValueError):
        idx.add(np.ones((1, 1000), dtype=np.float32))


def test_synthetic_fill_is_shard_invariant(gpu, oracle):
    """Philox rows keyed by global row id: two half-shards regenerate the same rows as one
    index (SURVEY §8d cfg 2/4), rows are unit norm, and search agrees with the oracle run on
    the downloaded rows."""
    from rassengine_amd.engine import Engine
    eng = Engine(0, 1024)
    try:
        whole = eng.open_index("syn-whole")
        whole.fill_synthetic(1000, seed=1234)
        lo = eng.open_index("syn-lo")
        lo.fill_synthetic(500, seed=1234, row_id_base=0)
        hi = eng.open_index("syn-hi")
        hi.fill_synthetic(500, seed=1234, row_id_base=500)
        rows = np.stack([whole.get_row(r) for r in range(0, 1000, 37)])
        np.testing.assert_allclose(np.linalg.norm(rows.astype(np.float64), axis=1), 1.0, atol=1e-6)
        for r in (0, 37, 499):
            assert np.array_equal(whole.get_row(r), lo.get_row(r))
        for r in (500, 501, 999):
            assert np.array_equal(whole.get_row(r), hi.get_row(r - 500))
        x = np.stack([whole.get_row(r) for r in range(1000)])
        q = np.random.default_rng(1).standard_normal((4, 1024), dtype=np.float32)
        s, i = whole.search(q, 10)
        rs, ri = oracle.search(x, oracle.normalize_ref(q).astype(np.float32), 10)
        assert np.array_equal(i, ri)
    finally:
        eng.close()
