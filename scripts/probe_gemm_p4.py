"""A/B of the persistent GEMMs: p5 (the product's) vs p4 (RASS_GEMM_VARIANT=p4, four waves of 128 x 128, deferred stores) on the
encoder's four shapes at M = 131 072: bit-equality of the outputs and us per launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rassengine_amd import _native as N
L = N.lib()
g = torch.Generator(device="cuda"); g.manual_seed(0)
M = int(os.environ.get("P4_M", 131072))
stream = ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))
def timed(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (Nn, K, epi, name) in [(3072, 1024, 0, "QKV"), (1024, 1024, 1, "attn-out"), (4096, 1024, 2, "FFN-up"), (1024, 4096, 1, "FFN-down")]:
    X = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((Nn, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn((Nn,), generator=g, device="cuda")
    R = torch.randn((M, Nn), generator=g, device="cuda").bfloat16()
    outs, times = {}, {}
    for variant in ("p5", "p4", "p5", "p4"):
        os.environ["RASS_GEMM_VARIANT"] = variant
        Y = torch.full((M, Nn), float("nan"), dtype=torch.bfloat16, device="cuda")
        def run():
            N.check("g", L.rass_gemm_bf16(ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                    ctypes.c_void_p(R.data_ptr()), ctypes.c_void_p(Y.data_ptr()), M, M, Nn, K, epi, stream))
        t = timed(run)
        outs[variant] = Y
        times.setdefault(variant, []).append(t)
    same = torch.equal(outs["p5"].view(torch.int16), outs["p4"].view(torch.int16))
    bad = int((outs["p5"].view(torch.int16) != outs["p4"].view(torch.int16)).sum())
    nan = int(torch.isnan(outs["p4"].float()).sum())
    fl = 2 * M * Nn * K
    print(f"{name:9s} N={Nn} K={K}: p5 {min(times['p5']):7.1f} us ({fl / min(times['p5']) / 1e6:.0f} TF/s)  p4 {min(times['p4']):7.1f} us "
          f"({fl / min(times['p4']) / 1e6:.0f} TF/s)  bit-equal {same} (mismatches {bad}, NaNs {nan})", flush=True)
