"""Fixed cost of the fused scan on SMALL slabs (the IVF's coarse scan is one: nlist = 4 096 centroids): us per search_device call
(normalise + scan + merge) at a few sizes, 32 and 16 queries.  Under rocprofv3 --kernel-trace the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rassengine_amd.engine import Engine, HipTimer
eng = Engine(0, 1024)
dev = torch.device("cuda", 0)
q = torch.randn((32, 1024), device=dev)
os_ = torch.empty((32, 32), device=dev); oi = torch.empty((32, 32), dtype=torch.int64, device=dev)
for n in (32, 1024, 4096, 8192, 65536):
    idx = eng.open_index(f"s{n}", capacity_rows=n)
    idx.fill_synthetic(n, seed=1)
    for nq, k in ((32, 8), (32, 1), (16, 8)):
        for _ in range(5): idx.search_device(q.data_ptr(), nq, k, os_.data_ptr(), oi.data_ptr())
        eng.synchronize()
        tm = HipTimer(); tm.start(eng.stream)
        for _ in range(100): idx.search_device(q.data_ptr(), nq, k, os_.data_ptr(), oi.data_ptr())
        tm.stop(eng.stream)
        print(f"rows {n:6d}  nq {nq:2d}  k {k:2d}: {tm.elapsed_ms() * 10:.1f} us per call", flush=True)
