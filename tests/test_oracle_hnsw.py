"""CPU tests of the HNSW restatement (oracle/hnsw.c): the reference's approximate index
(m / ef_construction of app/main.py:563-572) must approach the exact oracle as ef_search grows,
return true cosines for the ids it reports, and be deterministic for a seed."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def graph(oracle):
    rng = np.random.default_rng(11)
    n, d = 3000, 64
    centres = rng.standard_normal((40, d)).astype(np.float32)
    x = centres[rng.integers(0, 40, size=n)] + 0.7 * rng.standard_normal((n, d)).astype(np.float32)
    xn = oracle.normalize_ref(x).astype(np.float32)
    q = oracle.normalize_ref(centres[rng.integers(0, 40, size=64)]
                             + 0.7 * rng.standard_normal((64, d)).astype(np.float32)).astype(np.float32)
    h = oracle.Hnsw(xn, m=16, ef_construction=100, seed=5)
    yield oracle, xn, q, h
    h.close()


def _recall(a, b):
    return float(np.mean([len(set(a[r]) & set(b[r])) / a.shape[1] for r in range(a.shape[0])]))


def test_recall_approaches_exact(graph):
    O, xn, q, h = graph
    _, ri = O.search(xn, q, 5)
    rec = [_recall(h.search(q, 5, ef)[1], ri) for ef in (5, 20, 80, 320)]
    assert all(b >= a - 0.02 for a, b in zip(rec, rec[1:])), rec
    assert rec[-1] >= 0.99, rec
    evals = [h.search(q, 5, ef)[2] for ef in (5, 80, 320)]
    assert evals[0] < evals[1] < evals[2] <= q.shape[0] * (xn.shape[0] + 64)


def test_scores_are_true_cosines_and_sorted(graph):
    O, xn, q, h = graph
    s, i, _ = h.search(q, 10, 100)
    assert np.all(i >= 0) and np.all(np.diff(s, axis=1) <= 0)
    for r in range(q.shape[0]):
        assert len(set(i[r])) == 10
        true = xn[i[r]].astype(np.float64) @ q[r].astype(np.float64)
        assert np.max(np.abs(true - s[r])) <= 1e-5   # fp32 dot of 64 terms, tolerance stated


def test_deterministic_and_thread_independent(graph):
    O, xn, q, h = graph
    a = h.search(q, 5, 50, threads=1)
    b = h.search(q, 5, 50, threads=4)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    h2 = O.Hnsw(xn, m=16, ef_construction=100, seed=5)
    try:
        c = h2.search(q, 5, 50, threads=2)
        assert np.array_equal(a[1], c[1])
        assert h2.build_distance_evals == h.build_distance_evals
    finally:
        h2.close()


def test_tiny_and_bad_arguments(oracle):
    xn = oracle.normalize_ref(np.eye(4, 8, dtype=np.float32)).astype(np.float32)
    h = oracle.Hnsw(xn, m=4, ef_construction=8)
    s, i, _ = h.search(xn, 6, 16)          # k > rows: padded with -1 / -inf
    assert np.array_equal(i[:, 0], np.arange(4)) and np.all(i[:, 4:] == -1) and np.all(np.isneginf(s[:, 4:]))
    h.close()
    with pytest.raises(ValueError):
        oracle.Hnsw(xn, m=1)
