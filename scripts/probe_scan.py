"""Quick perf probe of the flat scan on one GPU (not the contract bench: see bench.py)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rassengine_amd.engine import Engine, HipTimer, scan_kernel_name

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=1024)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--batches", type=str, default="1,8,16,32")
a = ap.parse_args()

torch.cuda.init()
eng = Engine(0, a.dim)
idx = eng.open_index("probe", a.n)
idx.fill_synthetic(a.n, 1234)
eng.synchronize()
stream = eng.stream
bytes_scan = a.n * idx.row_stride * 4
for b in [int(x) for x in a.batches.split(",")]:
    q = torch.randn((b, a.dim), device="cuda")
    os_ = torch.empty((b, a.k), device="cuda")
    oi = torch.empty((b, a.k), dtype=torch.int64, device="cuda")
    for _ in range(3):
        idx.search_device(q.data_ptr(), b, a.k, os_.data_ptr(), oi.data_ptr())
    torch.cuda.synchronize()
    t = HipTimer()
    t.start(stream)
    for _ in range(a.iters):
        idx.search_device(q.data_ptr(), b, a.k, os_.data_ptr(), oi.data_ptr())
    t.stop(stream)
    ms = t.elapsed_ms() / a.iters
    print(f"B={b:3d} {ms*1e3:9.1f} us/scan  {bytes_scan/ms/1e6:8.1f} GB/s  {100*bytes_scan/ms/1e6/8000:5.1f}% of 8TB/s  "
          f"qps={b/ms*1e3:10.0f}  kernel={scan_kernel_name(a.dim,b)}", flush=True)
