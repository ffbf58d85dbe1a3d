"""Calibration only: the encoder GEMM shapes through this repo's ring kernel vs torch.matmul
(hipBLASLt/rocBLAS) on the same operands — what a tuned library reaches on this box.  The library
is NOT on the product path; this tells us how much headroom the hand-written kernel has left."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from rassengine_amd import _native as N
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=131072)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
L = N.lib()
g = torch.Generator(device="cuda"); g.manual_seed(0)
M = a.m
stream = ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))
def timed(fn):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e-3
for (Nn, K, epi) in [(3072, 1024, 0), (1024, 1024, 1), (4096, 1024, 2), (1024, 4096, 1)]:
    X = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((Nn, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn((Nn,), generator=g, device="cuda")
    bb = b.bfloat16()
    R = torch.randn((M, Nn), generator=g, device="cuda").bfloat16()
    Y = torch.empty((M, Nn), dtype=torch.bfloat16, device="cuda")
    def ours():
        N.check("g", L.rass_gemm_bf16(ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                ctypes.c_void_p(R.data_ptr()), ctypes.c_void_p(Y.data_ptr()), M, M, Nn, K, epi, stream))
    def blas_plain():
        torch.matmul(X, W.t(), out=Y)
    def blas_bias():
        F.linear(X, W, bb)
    fl = 2 * M * Nn * K
    t1, t2, t3 = timed(ours), timed(blas_plain), timed(blas_bias)
    print(f"M={M} N={Nn} K={K}: ours(epi={epi}) {t1*1e6:.0f} us {fl/t1/1e12:.0f} TF/s | matmul {t2*1e6:.0f} us {fl/t2/1e12:.0f} TF/s | "
          f"linear+bias {t3*1e6:.0f} us {fl/t3/1e12:.0f} TF/s", flush=True)
