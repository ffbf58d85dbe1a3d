"""Concurrency stress of the index through the C ABI: one thread appends rows (forcing several
geometric re-allocations of the HBM slab) and tombstones some, two threads search all the while.
Every intermediate answer must be well-formed (ids below the row count at return time, scores in
[-1, 1], best-first) and the final state must equal the oracle's answer over the final rows."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_add_delete_search(gpu, oracle):
    from rassengine_amd.engine import Engine
    rng = np.random.default_rng(77)
    dim, total, batch = 256, 24000, 300
    x = rng.standard_normal((total, dim)).astype(np.float32)
    xn = oracle.normalize_ref(x).astype(np.float32)
    q = rng.standard_normal((6, dim)).astype(np.float32)
    qn = oracle.normalize_ref(q).astype(np.float32)
    eng = Engine(0, dim)
    errors = []
    deleted = set()
    try:
        idx = eng.open_index("stress", capacity_rows=64)   # tiny: the slab must grow ~9 times
        idx.add(x[:batch])
        stop = threading.Event()

        def writer():
            try:
                r = np.random.default_rng(1)
                for lo in range(batch, total, batch):
                    idx.add(x[lo:lo + batch])
                    if lo % (4 * batch) == 0:
                        victim = int(r.integers(0, lo))
                        idx.delete(victim)
                        deleted.add(victim)
            except Exception as e:  # noqa: BLE001
                errors.append(("writer", repr(e)))
            finally:
                stop.set()

        def reader(seed):
            try:
                n = 0
                while not stop.is_set() or n < 5:
                    s, i = idx.search(q, 10)
                    rows_after = idx.rows
                    assert i.shape == (6, 10) and np.all(i < rows_after), (i.max(), rows_after)
                    live = i >= 0
                    assert np.all(s[live] <= 1.0 + 1e-5) and np.all(s[live] >= -1.0 - 1e-5)
                    assert np.all(np.diff(s, axis=1)[live[:, 1:]] <= 0)
                    n += 1
            except Exception as e:  # noqa: BLE001
                errors.append((f"reader{seed}", repr(e)))

        threads = [threading.Thread(target=writer)] + [threading.Thread(target=reader, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in threads), "stress threads did not finish"
        assert not errors, errors
        assert idx.rows == total and idx.count == total - len(deleted)
        s, i = idx.search(q, 10)
        tags = np.zeros(total, dtype=np.int32)
        tags[list(deleted)] = -1
        rs, ri = oracle.search(xn, qn, 10, tags=tags)
        assert np.array_equal(i, ri)
        assert np.max(np.abs(s.astype(np.float64) - rs)) <= 2e-6
    finally:
        eng.close()


def test_concurrent_adds_to_different_indices_share_the_engine_staging_buffer(gpu):
    """Two users ingesting at once (one engine, two indices): the host -> device staging buffer is
    engine scratch, so each chunk's upload + pack must be atomic with respect to the other index."""
    from rassengine_amd.engine import Engine
    dim, n, batch = 128, 20000, 37          # many small uploads to maximise interleaving
    eng = Engine(0, dim)
    errors = []
    try:
        data = {}
        for name, seed in (("user-a", 1), ("user-b", 2)):
            rng = np.random.default_rng(seed)
            data[name] = rng.standard_normal((n, dim)).astype(np.float32)
        idxs = {name: eng.open_index(name) for name in data}

        def ingest(name):
            try:
                for lo in range(0, n, batch):
                    idxs[name].add(data[name][lo:lo + batch], normalize=False)
            except Exception as e:  # noqa: BLE001
                errors.append((name, repr(e)))

        threads = [threading.Thread(target=ingest, args=(name,)) for name in data]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not errors, errors
        for name in data:
            assert idxs[name].rows == n
            assert np.array_equal(idxs[name].get_rows(0, n), data[name]), name
    finally:
        eng.close()
