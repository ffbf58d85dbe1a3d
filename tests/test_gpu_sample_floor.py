"""The scan's sample floor (api.hip scan_launch, ScanArgs::floors): a pre-pass over the first tile pair of every
workgroup gives each query a score that k rows are known to reach, and the big scan skips rows below it.  It must
never change a result: same ids and bit-identical scores with it on, off and forced, against the CPU oracle, with
filters whose matches are rare in the sample, tombstones, duplicated rows (ties with the floor) and k > 32 passes.
RASS_SCAN_SAMPLE_FLOOR is read at every launch, so one process can compare the settings."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_F64 = 2e-6


class _Floor:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = os.environ.get("RASS_SCAN_SAMPLE_FLOOR")
        os.environ["RASS_SCAN_SAMPLE_FLOOR"] = self.mode

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("RASS_SCAN_SAMPLE_FLOOR", None)
        else:
            os.environ["RASS_SCAN_SAMPLE_FLOOR"] = self.old


def _swaps_are_ties(i_gpu, i_ref, all64):
    for q in range(i_ref.shape[0]):
        for a, b in zip(i_gpu[q], i_ref[q]):
            if a != b and (a < 0 or b < 0 or abs(all64[q, a] - all64[q, b]) > 2 * TOL_F64):
                return False
    return True


@pytest.fixture(scope="module")
def corpus(gpu):
    """40,000 rows: more than twice the 16,384-row sample of a 256-CU part, small enough for the oracle."""
    from rassengine_amd.engine import Engine
    rng = np.random.default_rng(77)
    n, dim = 40_000, 1024
    x = rng.standard_normal((n, dim)).astype(np.float32)
    # rows 30,000.. duplicate sample rows 100..: every score the sample reaches is reached again outside it
    x[30_000:30_400] = x[100:500]
    patient = rng.integers(1, 2000, size=n).astype(np.int32)   # ~20 rows per patient: rare in the sample
    patient[:64] = 5                                           # one patient the sample knows well
    patient[20_000:20_040] = 6                                 # one it never sees
    doctype = rng.integers(1, 3, size=n).astype(np.int32)
    tags = (patient | (doctype << 24)).astype(np.int32)
    eng = Engine(0, dim)
    ix = eng.open_index("floor")
    ix.add(x, tags=tags)
    dead = rng.choice(n, 500, replace=False)
    for r in dead[:50]:
        ix.delete(int(r))
    tags_live = tags.copy()
    tags_live[dead[:50]] = -1
    yield ix, x, tags_live
    eng.close()


def _three_ways(ix, q, k, **kw):
    out = {}
    for mode in ("0", "force", "1"):
        with _Floor(mode):
            out[mode] = ix.search(q, k, **kw)
    for mode in ("force", "1"):
        assert np.array_equal(out[mode][1], out["0"][1]), mode
        assert np.array_equal(out[mode][0].view(np.uint32), out["0"][0].view(np.uint32)), mode
    return out["force"]


def test_unfiltered_batch_matches_oracle(corpus, oracle):
    ix, x, tags = corpus
    rng = np.random.default_rng(5)
    q = rng.standard_normal((32, 1024)).astype(np.float32)
    q[3] = x[120] * 3.0      # its best rows are a sample row and that row's duplicate outside the sample
    q[4] = x[39_999]
    s, i = _three_ways(ix, q, 10)
    xn = oracle.normalize_ref(x).astype(np.float32)
    qn = oracle.normalize_ref(q).astype(np.float32)
    rs, ri = oracle.search(xn, qn, 10, tags=np.where(tags == -1, -1, 0).astype(np.int32))
    all64 = oracle.scores(xn, qn)
    assert _swaps_are_ties(i, ri, all64)
    assert np.all(np.abs(s.astype(np.float64) - rs) <= TOL_F64)
    assert list(i[3, :2]) == [120, 30_020] and s[3, 0] == s[3, 1]
    assert i[4, 0] == 39_999


def test_filtered_batch_matches_oracle(corpus, oracle):
    """Per-query patient filters: most patients have fewer than k rows in the sample (no floor), patient 5 has
    64 of them there, patient 6 none, 99999 matches nothing anywhere."""
    ix, x, tags = corpus
    rng = np.random.default_rng(6)
    q = rng.standard_normal((32, 1024)).astype(np.float32)
    PM = 0x00FFFFFF
    qf = rng.integers(1, 2000, size=32).astype(np.int32)
    qf[0], qf[1], qf[2], qf[3] = 5, 6, 99_999, -1
    qm = np.full(32, PM, dtype=np.int32)
    s, i = _three_ways(ix, q, 10, q_filter=qf, q_filter_mask=qm)
    xn = oracle.normalize_ref(x).astype(np.float32)
    qn = oracle.normalize_ref(q).astype(np.float32)
    rs, ri = oracle.search(xn, qn, 10, tags=tags, qfilter=qf, qmask=qm)
    all64 = oracle.scores(xn, qn)
    assert _swaps_are_ties(i, ri, all64)
    assert np.array_equal(i >= 0, ri >= 0)
    valid = ri >= 0
    assert np.all(np.abs(s[valid].astype(np.float64) - rs[valid]) <= TOL_F64)
    assert np.all(i[2] == -1)
    assert np.all((tags[i[0]] & PM) == 5) and np.all((tags[i[1]] & PM) == 6)


@pytest.mark.parametrize("k", [1, 32, 70])
def test_other_k_incl_passes_beyond_32(corpus, oracle, k):
    """k = 70 runs three continuation passes, each with its own sample floor under its own bound."""
    ix, x, tags = corpus
    rng = np.random.default_rng(8 + k)
    q = rng.standard_normal((20, 1024)).astype(np.float32)
    s, i = _three_ways(ix, q, k)
    xn = oracle.normalize_ref(x).astype(np.float32)
    qn = oracle.normalize_ref(q).astype(np.float32)
    rs, ri = oracle.search(xn, qn, k, tags=np.where(tags == -1, -1, 0).astype(np.int32))
    all64 = oracle.scores(xn, qn)
    assert _swaps_are_ties(i, ri, all64)
    assert np.all(np.abs(s.astype(np.float64) - rs) <= TOL_F64)


def test_full_size_on_equals_off(gpu):
    """1M x 1024, 32 queries (the bench's launch group): on and off agree bit for bit."""
    from rassengine_amd.engine import Engine
    eng = Engine(0, 1024)
    try:
        ix = eng.open_index("full", capacity_rows=1_000_000)
        ix.fill_synthetic(1_000_000, seed=99)
        eng.synchronize()
        rng = np.random.default_rng(1)
        q = rng.standard_normal((32, 1024)).astype(np.float32)
        q[0] = ix.get_row(999_999)
        with _Floor("0"):
            s0, i0 = ix.search(q, 10)
        with _Floor("1"):
            s1, i1 = ix.search(q, 10)
        assert np.array_equal(i0, i1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32))
        assert i1[0, 0] == 999_999
    finally:
        eng.close()
