"""(Lives under tests/: it times the CPU oracle, which only tests/, smoke() and bench.py's cpu_baseline leg may touch.)

BASELINE.json configs[0] — the reference's own CPU-runnable case: one user's index of
10 000 x 1024 unit vectors, top_k = 5, one query per request (app/main.py:1093-1107, 1552).

GPU side: the HIP flat index through the C ABI (host query in, host top-5 out: what one /ask
pays, PCIe included) and device-resident batches.  CPU side: the HNSW restatement with the
reference's parameters (oracle/hnsw.c: m 48, ef_construction 400, ef_search 512) and the exact
flat scan, on this box's host cores.  Recall@5 of HNSW vs the exact top-5 is reported next to
the HIP path's (1.0 by construction, ids compared with the f64 oracle)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from rassengine_amd.engine import Engine, HipTimer
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000)
ap.add_argument("--queries", type=int, default=1000)
ap.add_argument("--k", type=int, default=5)
ap.add_argument("--ef-search", type=int, default=O.Hnsw.EF_SEARCH)
a = ap.parse_args()
dim, k = 1024, a.k

eng = Engine(0, dim)
idx = eng.open_index("cfg1", capacity_rows=a.rows)
idx.fill_synthetic(a.rows, seed=1234)
eng.synchronize()
x = idx.get_rows(0, a.rows)                       # the same rows for the CPU side
rng = np.random.default_rng(3)
# queries near corpus rows (a /ask query resembles some chunk) + pure noise queries, half each
near = x[rng.integers(0, a.rows, size=a.queries // 2)] + 0.5 * rng.standard_normal((a.queries // 2, dim)).astype(np.float32) / np.sqrt(dim)
q = np.concatenate([near, rng.standard_normal((a.queries - a.queries // 2, dim)).astype(np.float32)])
qn = O.normalize_ref(q).astype(np.float32)

# --- GPU: one query per call, host buffers (PCIe + launch + sync inclusive)
for i in range(20):
    idx.search(qn[i:i + 1], k)
lat, ids_gpu = [], np.empty((a.queries, k), dtype=np.int64)
for i in range(a.queries):
    t0 = time.perf_counter()
    s, ids = idx.search(qn[i:i + 1], k)
    lat.append(time.perf_counter() - t0)
    ids_gpu[i] = ids[0]
lat = np.array(lat) * 1e6
# --- GPU: device-resident queries, batch 1 and 32 (kernel-side rate)
dq = torch.from_numpy(qn).cuda()
out_s = torch.empty((32, k), device="cuda"); out_i = torch.empty((32, k), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
tm = HipTimer()
dev = {}
for B in (1, 32):
    nb = (a.queries // B)
    for b in range(min(nb, 8)):
        idx.search_device(dq[b * B:(b + 1) * B].data_ptr(), B, k, out_s.data_ptr(), out_i.data_ptr())
    eng.synchronize()
    tm.start(eng.stream)
    for b in range(nb):
        idx.search_device(dq[b * B:(b + 1) * B].data_ptr(), B, k, out_s.data_ptr(), out_i.data_ptr())
    tm.stop(eng.stream)
    dev[B] = nb * B / tm.elapsed_ms() * 1e3

# --- truth (f64 oracle) and the CPU paths
_, truth = O.search(x, qn, k)
rec = lambda ids: float(np.mean([len(set(ids[r]) & set(truth[r])) / k for r in range(a.queries)]))
t0 = time.perf_counter()
h = O.Hnsw(x)
build_s = time.perf_counter() - t0
t0 = time.perf_counter(); _, ids_h, ev = h.search(qn, k, a.ef_search, threads=1); t1 = time.perf_counter() - t0
t0 = time.perf_counter(); h.search(qn, k, a.ef_search, threads=0); tall = time.perf_counter() - t0
sweep = []
for ef in (16, 64, 128, 256, 512):
    t0 = time.perf_counter(); _, i_e, ev_e = h.search(qn, k, ef, threads=1); dt = time.perf_counter() - t0
    sweep.append({"ef_search": ef, "recall_at_5": round(rec(i_e), 4), "qps_1thread": round(a.queries / dt, 1),
                  "dist_evals_per_query": round(ev_e / a.queries, 1)})
t0 = time.perf_counter(); O.search(x, qn[:200], k, kind=O.KIND_F32_FAST, threads=1); tflat1 = time.perf_counter() - t0
print(json.dumps({
    "workload": f"cfg 1: {a.rows} x {dim} fp32 rows, top-{k}, one query per request",
    "hip_host_api": {"p50_us": round(float(np.percentile(lat, 50)), 1), "p99_us": round(float(np.percentile(lat, 99)), 1),
                     "qps_serial": round(1e6 / float(lat.mean()), 1), "recall_at_5": rec(ids_gpu)},
    "hip_device_resident_qps": {"B1": round(dev[1], 1), "B32": round(dev[32], 1)},
    "cpu_hnsw": {"m": O.Hnsw.M, "ef_construction": O.Hnsw.EF_CONSTRUCTION, "ef_search": a.ef_search,
                 "build_s": round(build_s, 1), "build_dist_evals": h.build_distance_evals,
                 "qps_1thread": round(a.queries / t1, 1), "qps_all_threads": round(a.queries / tall, 1),
                 "threads": O.num_threads(), "recall_at_5": round(rec(ids_h), 4),
                 "dist_evals_per_query": round(ev / a.queries, 1), "sweep": sweep},
    "cpu_flat_exact_qps_1thread": round(200 / tflat1, 1),
}))
