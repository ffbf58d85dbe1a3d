"""Launch-to-launch time of the few-rows GEMMs at a query's shapes (rass_gemm_bf16_ws with a scratch: the encoder's query-time path),
back to back on one stream: prices the variants of the one-query forward (profiles/r04_query_forward_fusion.txt, section 8)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rassengine_amd import _native as N
L = N.lib()
def run(M, Nn, K, epi, iters=2000):
    x = torch.randn((128, K), device="cuda").bfloat16()
    w = (torch.randn((Nn, K), device="cuda") * 0.03).bfloat16()
    b = torch.randn((Nn,), device="cuda")
    r = torch.randn((128, Nn), device="cuda").bfloat16()
    y = torch.empty((128, Nn), device="cuda", dtype=torch.bfloat16)
    ws = torch.empty((4 * 128 * 4096,), device="cuda")
    st = ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))
    def go():
        N.check("g", L.rass_gemm_bf16_ws(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                         ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(y.data_ptr()), M, 128, Nn, K, epi,
                                         ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4, st))
    for _ in range(20): go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): go()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
for (M, Nn, K, epi, name) in [(12, 3072, 1024, 0, "QKV 4-wave"), (12, 1024, 4096, 1, "FFN-down 16-wave EPI1"), (12, 1024, 1024, 1, "attn-out 4-wave"),
                              (12, 4096, 1024, 2, "FFN-up 4-wave GELU"), (24, 1024, 4096, 1, "FFN-down 16-wave 24 rows")]:
    print(f"{name:28s} M={M:3d}: {run(M, Nn, K, epi):6.2f} us per launch (back to back, each re-reading its weights from L2/HBM)")
