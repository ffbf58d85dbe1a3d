"""Prefilter mode: bf16 candidate scan + exact fp32 re-rank.  Returned scores
must be BIT-IDENTICAL to the flat fp32 path for every returned row; on these (random,
well-separated) corpora the id lists must be identical too (recall 1.0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def engine(gpu):
    from rassengine_amd.engine import Engine
    eng = Engine(device=0, dim=1024)
    yield eng
    eng.close()


def test_prefilter_matches_flat_path(engine, oracle):
    rng = np.random.default_rng(8)
    idx = engine.open_index("pf")
    idx.set_prefilter(True)                      # enabled before any row exists
    assert idx.prefilter
    n = 0
    tags_all = []
    for c in (1, 40, 3000, 17, 9000):            # odd batch sizes: bf16 slab kept in sync on add + growth
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
        idx.set_prefilter(True)                  # re-enabled on a populated index: converts existing rows
        assert np.array_equal(i_p, i_f), k
        assert np.array_equal(s_p, s_f), k       # exact re-rank = the flat kernel's fmaf order
    live = i_p[i_p >= 0]
    assert 7 not in live and 4000 not in live
    for r in range(45):
        if qf[r] >= 0:
            assert np.all(tags[i_p[r][i_p[r] >= 0]] == qf[r])


def test_prefilter_small_and_padding(engine):
    rng = np.random.default_rng(9)
    idx = engine.open_index("pf-small")
    x = rng.standard_normal((5, 1024), dtype=np.float32)
    idx.add(x)
    idx.set_prefilter(True)
    s, i = idx.search(x[:2], 10)
    idx.set_prefilter(False)
    s2, i2 = idx.search(x[:2], 10)
    assert np.array_equal(i, i2) and np.array_equal(s, s2)
    assert np.all(i[:, 5:] == -1) and np.all(np.isneginf(s[:, 5:]))
    assert i[0, 0] == 0 and i[1, 0] == 1


def test_prefilter_near_duplicates_are_reranked_exactly(engine, oracle):
    """Rows closer together than bf16 resolution: the candidate scan cannot order them, the
    fp32 re-rank must (and equal scores must come back id-ascending)."""
    rng = np.random.default_rng(10)
    base = rng.standard_normal((1, 1024)).astype(np.float32)
    x = np.repeat(base, 20, axis=0) + 1e-4 * rng.standard_normal((20, 1024)).astype(np.float32)
    x[11] = x[3]                                  # exact duplicate
    filler = rng.standard_normal((2000, 1024)).astype(np.float32)
    idx = engine.open_index("pf-dup")
    idx.add(np.concatenate([filler, x]))
    q = base * 3.0
    for k in (12, 16, 20):                        # 12, 16: prefilter path; 20: falls back to the flat scan
        idx.set_prefilter(True)
        s_p, i_p = idx.search(q, k)
        idx.set_prefilter(False)
        s_f, i_f = idx.search(q, k)
        assert np.array_equal(i_p, i_f) and np.array_equal(s_p, s_f), k
        assert set(i_f[0]) <= set(range(2000, 2020))
        # many of these rows tie EXACTLY in fp32 (|1 - cos| ~ 1e-8 < ulp): non-increasing scores and,
        # inside every tie group, ascending ids
        for a, b, sa, sb in zip(i_f[0][:-1], i_f[0][1:], s_f[0][:-1], s_f[0][1:]):
            assert sa > sb or (sa == sb and a < b)
    pos = list(i_f[0])
    assert s_f[0][pos.index(2003)] == s_f[0][pos.index(2011)] and pos.index(2003) < pos.index(2011)


def test_prefilter_needs_stride_multiple_of_256(gpu):
    from rassengine_amd.engine import Engine
    from rassengine_amd._native import RassError
    eng = Engine(0, 384)
    try:
        idx = eng.open_index("pf-384")
        with pytest.raises(RassError):
            idx.set_prefilter(True)
    finally:
        eng.close()
