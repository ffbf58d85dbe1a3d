"""A/B timing of the flat scan step across builds of librass_hip (RASS_HIP_LIB) — kernel experiments."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rassengine_amd.engine import Engine, HipTimer
n, dim, k = int(os.environ.get("N", 1_000_000)), 1024, 10
eng = Engine(0, dim)
idx = eng.open_index("probe", n)
idx.fill_synthetic(n, 1234)
eng.synchronize()
out = []
for b in (32, 16):
    q = torch.randn((b, dim), device="cuda")
    os_ = torch.empty((b, k), device="cuda"); oi = torch.empty((b, k), dtype=torch.int64, device="cuda")
    for _ in range(60):
        idx.search_device(q.data_ptr(), b, k, os_.data_ptr(), oi.data_ptr())
    eng.synchronize()
    eng.kernel_timing_begin(200)
    t0 = time.perf_counter()
    for _ in range(200):
        idx.search_device(q.data_ptr(), b, k, os_.data_ptr(), oi.data_ptr())
    eng.synchronize()
    wall = (time.perf_counter() - t0) / 200
    ms, launches = eng.kernel_timing_end()
    out.append(f"B={b}: scan kernel {ms / launches * 1e3:7.1f} us, whole search {wall * 1e6:7.1f} us")
print(os.path.basename(os.environ.get("RASS_HIP_LIB", "default")), "sample_floor=" + os.environ.get("RASS_SCAN_SAMPLE_FLOOR", "1"), "  ".join(out), flush=True)
