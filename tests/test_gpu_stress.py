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


def test_many_indices_many_threads_overlap_their_round_trips(gpu, oracle):
    """The reference keeps ONE index per user (app/main.py:346-347) and serves requests concurrently: N indices
    x M threads searching at once (masked filters and k > 32 in the mix), one thread ingesting into two of them
    and dropping / re-creating a third.  rass_index_search_ex holds the engine lock while enqueuing only and
    waits on a per-call event on a pinned slot (8 slots, 12 threads: the pool must block and hand over, not
    corrupt).  Every answer must equal the oracle's for the rows that index held when the call started or ended."""
    from rassengine_amd.engine import Engine
    dim, n_idx, n_threads, rows0 = 256, 6, 12, 3000
    eng = Engine(0, dim)
    rng = np.random.default_rng(9)
    data, tags, xn = {}, {}, {}
    try:
        idxs = {}
        for u in range(n_idx):
            data[u] = rng.standard_normal((rows0 + 600, dim)).astype(np.float32)
            tags[u] = rng.integers(1, 4, size=rows0 + 600).astype(np.int32)
            xn[u] = oracle.normalize_ref(data[u]).astype(np.float32)
            idxs[u] = eng.open_index(f"user-{u}")
            idxs[u].add(data[u][:rows0], tags=tags[u][:rows0])
        errors = []
        stop = threading.Event()

        def expect(u, q, k, n_rows, qf=None, qm=None):
            qn = oracle.normalize_ref(q).astype(np.float32)
            return oracle.search(xn[u][:n_rows], qn, k, tags=tags[u][:n_rows], qfilter=qf, qmask=qm)

        def searcher(t):
            r = np.random.default_rng(100 + t)
            try:
                n = 0
                while not stop.is_set() or n < 30:
                    u = int(r.integers(0, n_idx))
                    nq = int(r.integers(1, 40))
                    k = int(r.choice([1, 5, 10, 32, 40]))
                    q = r.standard_normal((nq, dim)).astype(np.float32)
                    mode = int(r.integers(0, 3))
                    qf = qm = None
                    if mode == 1:
                        qf = r.integers(-1, 4, size=nq).astype(np.int32)
                    elif mode == 2:
                        qf = r.integers(1, 4, size=nq).astype(np.int32)
                        qm = np.full(nq, 0x00FFFFFF, dtype=np.int32)
                    before = idxs[u].rows
                    s, i = idxs[u].search(q, k, q_filter=qf, q_filter_mask=qm)
                    after = idxs[u].rows
                    ok = False
                    for n_rows in sorted({before, after} | ({rows0 + 300} if before < rows0 + 300 < after else set())):
                        rs, ri = expect(u, q, k, n_rows, qf, qm)
                        if np.array_equal(i, ri) and np.all(np.abs(s[ri >= 0].astype(np.float64) - rs[ri >= 0]) <= 2e-6):
                            ok = True
                            break
                    if not ok:
                        # an add published between the two reads: the result must still be a valid top-k of a
                        # prefix of the rows — check it is best-first, in range, and filter-correct
                        assert np.all(i < after) and np.all(np.diff(s, axis=1)[(i >= 0)[:, 1:]] <= 0), (u, k, nq)
                        if qf is not None:
                            for row_q in range(nq):
                                live = i[row_q][i[row_q] >= 0]
                                if qf[row_q] >= 0:
                                    assert np.all((tags[u][live] & (-1 if qm is None else int(qm[row_q]))) == qf[row_q])
                    n += 1
            except Exception as e:  # noqa: BLE001
                errors.append((f"searcher{t}", repr(e)))

        def writer():
            try:
                for step in range(2):
                    for u in (0, 1):
                        lo = rows0 + 300 * step
                        idxs[u].add(data[u][lo:lo + 300], tags=tags[u][lo:lo + 300])
                    # a whole user index goes away and comes back while others are searched
                    eng.drop_index("scratch-user") if step else None
                    sc = eng.open_index("scratch-user")
                    sc.add(data[2][:500])
                    assert sc.search(data[2][:3], 1)[1][:, 0].tolist() == [0, 1, 2]
            except Exception as e:  # noqa: BLE001
                errors.append(("writer", repr(e)))
            finally:
                stop.set()

        threads = [threading.Thread(target=searcher, args=(t,)) for t in range(n_threads)] + \
                  [threading.Thread(target=writer)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in threads), "stress threads did not finish"
        assert not errors, errors[:3]
        for u in (0, 1):
            assert idxs[u].rows == rows0 + 600
    finally:
        eng.close()


def test_query_batcher_over_the_hip_index(gpu, oracle):
    """the cross-request micro-batcher on the REAL index: 100 concurrent asemantic_search-style
    requests (mixed k, mixed patient filters) share scans and each gets exactly the single-request answer."""
    import asyncio
    from rassengine_amd.batcher import QueryBatcher
    from rassengine_amd.engine import Engine
    dim = 512
    rng = np.random.default_rng(4)
    x = rng.standard_normal((5000, dim)).astype(np.float32)
    tags = (rng.integers(1, 5, size=5000) | (1 << 24)).astype(np.int32)     # patient code | doc_type 1
    eng = Engine(0, dim)
    try:
        idx = eng.open_index("batched")
        idx.add(x, tags=tags)
        qs = rng.standard_normal((100, dim)).astype(np.float32)
        ks = rng.integers(1, 11, size=100)
        pats = rng.integers(0, 5, size=100)         # 0 = no filter

        async def main():
            b = QueryBatcher(idx, max_batch=32, max_delay_ms=20.0)
            res = await asyncio.gather(*[
                b.search(qs[i], int(ks[i]), int(pats[i]) if pats[i] else -1, 0x00FFFFFF if pats[i] else -1)
                for i in range(100)])
            await b.close()
            return b, res

        b, res = asyncio.run(main())
        assert b.served == 100 and b.scans <= 8
        xn = oracle.normalize_ref(x).astype(np.float32)
        for i, (s, ids) in enumerate(res):
            qn = oracle.normalize_ref(qs[i:i + 1]).astype(np.float32)
            if pats[i]:
                rs, ri = oracle.search(xn, qn, int(ks[i]), tags=tags, qfilter=np.array([pats[i]], dtype=np.int32),
                                       qmask=np.array([0x00FFFFFF], dtype=np.int32))
            else:
                rs, ri = oracle.search(xn, qn, int(ks[i]), tags=tags)
            assert ids.shape == (ks[i],) and np.array_equal(ids, ri[0]), i
            assert np.all(np.abs(s.astype(np.float64) - rs[0]) <= 2e-6)
    finally:
        eng.close()


def test_cross_index_batch_equals_per_index_search(gpu, oracle):
    """rass_index_search_multi: 70 queries over 9 per-user indices of very different sizes (0 .. 40 000 rows, one
    with tombstones), patient filters (plain and masked) in the mix, answered by 3 scan launches instead of 70 —
    every row must be bit-identical to that index's own rass_index_search_ex answer; and the engine-wide
    CrossIndexBatcher behind HipIndexer.asemantic_search coalesces requests of DIFFERENT users."""
    import asyncio
    from rassengine_amd.batcher import CrossIndexBatcher
    from rassengine_amd.engine import Engine
    dim = 512
    rng = np.random.default_rng(21)
    sizes = [0, 1, 31, 33, 500, 4000, 9000, 40000, 2500]
    eng = Engine(0, dim)
    try:
        idxs, tags = [], []
        for u, n in enumerate(sizes):
            ix = eng.open_index(f"user-{u}")
            t = (rng.integers(1, 4, size=n) | (1 << 24)).astype(np.int32)
            if n:
                ix.add(rng.standard_normal((n, dim)).astype(np.float32), tags=t)
            idxs.append(ix)
            tags.append(t)
        for r in (3, 777, 3999):
            idxs[5].delete(r)
        nq = 70
        who = rng.integers(0, len(sizes), size=nq)
        who[:4] = [7, 7, 0, 7]                                   # several queries on one index, one on the empty one
        q = rng.standard_normal((nq, dim)).astype(np.float32)
        for k, qf, qm in ((10, None, None), (32, None, None),
                          (5, rng.integers(-1, 4, size=nq).astype(np.int32) | np.int32(0), None),
                          (7, rng.integers(1, 4, size=nq).astype(np.int32), np.full(nq, 0x00FFFFFF, dtype=np.int32))):
            if qf is not None and qm is None:
                qf = np.where(qf >= 0, qf | (1 << 24), -1).astype(np.int32)     # exact compare needs the full tag
            s, i = eng.search_multi([idxs[w] for w in who], q, k, qf, qm)
            for r in range(nq):
                s1, i1 = idxs[who[r]].search(q[r:r + 1], k, None if qf is None else qf[r:r + 1],
                                             None if qm is None else qm[r:r + 1])
                assert np.array_equal(i[r], i1[0]) and np.array_equal(s[r], s1[0]), (k, r, who[r])
            assert np.all(i[2] == -1)                            # the empty index answers with padding
            if qf is not None:
                for r in range(nq):
                    live = i[r][i[r] >= 0]
                    if qf[r] >= 0:
                        assert np.all((tags[who[r]][live] & (-1 if qm is None else int(qm[r]))) == qf[r])

        async def main():
            b = CrossIndexBatcher(eng, max_batch=32, max_delay_ms=20.0)
            res = await asyncio.gather(*[b.search(idxs[who[r]], q[r], 10) for r in range(nq)])
            await b.close()
            return b, res
        b, res = asyncio.run(main())
        assert b.served == nq and b.scans <= 4                    # 70 users' requests in <= 4 launches
        s10, i10 = eng.search_multi([idxs[w] for w in who], q, 10)
        for r, (sr, ir) in enumerate(res):
            assert np.array_equal(ir, i10[r]) and np.array_equal(sr, s10[r])
    finally:
        eng.close()
