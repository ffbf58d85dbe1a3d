"""GEMM-only probe (encoder shapes) for rocprofv3 PMC passes."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rassengine_amd import _native as N
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=131072)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--grid", type=int, default=0, help="RASS_GEMM_GRID: cap on the persistent kernel's workgroups; rows scale with it (same tiles per workgroup)")
a = ap.parse_args()
L = N.lib()
g = torch.Generator(device="cuda"); g.manual_seed(0)
M = a.m
if a.grid:
    os.environ["RASS_GEMM_GRID"] = str(a.grid); os.environ["RASS_GEMM_VARIANT"] = "p5"
    M = a.m * a.grid // 256
stream = ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))
for (Nn, K, epi) in [(3072, 1024, 0), (1024, 1024, 1), (4096, 1024, 2), (1024, 4096, 1)]:
    X = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((Nn, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn((Nn,), generator=g, device="cuda")
    R = torch.randn((M, Nn), generator=g, device="cuda").bfloat16()
    Y = torch.empty((M, Nn), dtype=torch.bfloat16, device="cuda")
    def run():
        N.check("g", L.rass_gemm_bf16(ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                ctypes.c_void_p(R.data_ptr()), ctypes.c_void_p(Y.data_ptr()), M, M, Nn, K, epi, stream))
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(f"M={M} N={Nn} K={K} epi={epi}: {dt*1e6:.0f} us  {2*M*Nn*K/dt/1e12:.0f} TF/s", flush=True)
