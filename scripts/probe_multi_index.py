"""What sharing launches across per-user indices buys (host API, PCIe + sync inclusive): 32 users with one index
of N rows each (the reference's model: one index per user, app/main.py:346-347), one query per user."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rassengine_amd.engine import Engine
dim, users, k = 1024, 32, 5
eng = Engine(0, dim)
rng = np.random.default_rng(0)
out = []
for n in (1_000, 10_000, 50_000):
    idxs = []
    for u in range(users):
        ix = eng.open_index(f"u{n}-{u}", capacity_rows=n)
        ix.fill_synthetic(n, seed=u + 1)
        idxs.append(ix)
    eng.synchronize()
    q = rng.standard_normal((users, dim)).astype(np.float32)
    for _ in range(5):
        for u in range(users):
            idxs[u].search(q[u:u + 1], k)
        eng.search_multi(idxs, q, k)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        for u in range(users):
            idxs[u].search(q[u:u + 1], k)
    t_seq = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        s, i = eng.search_multi(idxs, q, k)
    t_multi = (time.perf_counter() - t0) / reps
    s1 = np.stack([idxs[u].search(q[u:u + 1], k)[1][0] for u in range(users)])
    assert np.array_equal(s1, i)
    out.append({"rows_per_user": n, "users": users, "one_search_per_user_ms": round(t_seq * 1e3, 3),
                "one_cross_index_batch_ms": round(t_multi * 1e3, 3), "speedup": round(t_seq / t_multi, 1),
                "queries_per_s_batched": round(users / t_multi, 0)})
    for u in range(users):
        eng.drop_index(f"u{n}-{u}")
print(json.dumps(out))
