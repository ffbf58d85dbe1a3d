"""Generates tests/golden/search_N4096_D1024_Q32_seed1234.npz and search_small_N512_D256.npz.

Run from the repo root:  python tests/golden/make_search_fixtures.py
The expected ids / scores come from the CPU oracle's fp64 ranking (oracle/rass_oracle.c,
cross-checked here against a plain numpy fp64 matmul + lexsort).  The reference holds no
golden vectors for this path (tests/test_main.py:26), so these pin OUR oracle, not the
reference's HNSW ("parity unpinned").

The 4096x1024 corpus is not stored (16 MB): it is regenerated from the seed and verified by
its sha256; the small fixture stores its rows.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    xn = O.synthetic_unit_rows(4096, 1024, 1234)
    rng = np.random.default_rng(4321)
    q_raw = (rng.standard_normal((32, 1024), dtype=np.float32) * 2.5).astype(np.float32)
    qn = O.normalize_ref(q_raw).astype(np.float32)
    out = {"q_raw": q_raw, "xn_sha256": np.frombuffer(hashlib.sha256(xn.tobytes()).digest(), dtype=np.uint8),
           "seed": np.int64(1234), "n": np.int64(4096), "dim": np.int64(1024)}
    for k in (3, 5, 10):
        s, i = O.search(xn, qn, k, kind=O.KIND_F64)
        s2, i2 = O.search_numpy(xn, qn, k)
        assert np.array_equal(i, i2) and np.abs(s - s2).max() < 1e-12
        out[f"ids_k{k}"] = i
        out[f"scores_k{k}"] = s
    np.savez_compressed(os.path.join(HERE, "search_N4096_D1024_Q32_seed1234.npz"), **out)

    rng = np.random.default_rng(99)
    xs = O.normalize_ref(rng.standard_normal((512, 256), dtype=np.float32)).astype(np.float32)
    xs[100] = xs[7]  # exact duplicate rows: tie rule
    xs[300] = xs[7]
    qs = rng.standard_normal((6, 256), dtype=np.float32)
    qs[0] = xs[7] * 4.0
    tags = rng.integers(0, 4, size=512).astype(np.int32)
    tags[[11, 12, 13]] = -1
    qf = np.array([-1, 0, 1, 2, 3, -1], dtype=np.int32)
    qsn = O.normalize_ref(qs).astype(np.float32)
    s, i = O.search(xs, qsn, 10, tags=tags, qfilter=qf, kind=O.KIND_F64)
    np.savez_compressed(os.path.join(HERE, "search_small_N512_D256.npz"), xn=xs, q_raw=qs, tags=tags, qfilter=qf,
                        ids_k10=i, scores_k10=s)
    print("wrote fixtures")


if __name__ == "__main__":
    main()
