"""Single-query latency of the host API (what one /ask sees): cfg 1 (10k rows, k=5) and 1M rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rassengine_amd.engine import Engine
eng = Engine(0, 1024)
rng = np.random.default_rng(0)
for n in (10_000, 100_000, 1_000_000):
    idx = eng.open_index(f"lat{n}", capacity_rows=n)
    idx.fill_synthetic(n, seed=1234)
    eng.synchronize()
    q = rng.standard_normal((1000, 1024), dtype=np.float32)
    for i in range(20): idx.search(q[i:i+1], 5)
    lat = []
    for i in range(1000):
        t0 = time.perf_counter(); idx.search(q[i:i+1], 5); lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e6
    print(f"N={n:8d} k=5 B=1 host API: p50 {np.percentile(lat,50):7.1f} us  p99 {np.percentile(lat,99):7.1f} us  -> {1e6/lat.mean():8.0f} qps serial", flush=True)
    eng.drop_index(f"lat{n}")

# PCIe-inclusive batch rate (host buffers in, host results out: what the C ABI's host entry costs)
n = 1_000_000
idx = eng.open_index("pcie", capacity_rows=n)
idx.fill_synthetic(n, seed=1234)
eng.synchronize()
q = rng.standard_normal((32 * 64, 1024), dtype=np.float32)
for i in range(5): idx.search(q[:32], 10)
t0 = time.perf_counter()
for i in range(64): idx.search(q[32 * i:32 * i + 32], 10)
dt = time.perf_counter() - t0
print(f"N={n:8d} k=10 B=32 host API (H2D 128 KB + D2H 3.8 KB + sync per batch): {32 * 64 / dt:8.0f} qps, {dt / 64 * 1e6:.1f} us per batch", flush=True)
