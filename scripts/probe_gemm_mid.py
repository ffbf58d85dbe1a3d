"""Sweep of the split-K slice count of the 128^2 GEMM kernels on the encoder's four GEMM shapes at the row counts the
embed micro-batcher and small uploads produce (65 .. ~3 000 rows).  Time = mean of back-to-back launches on one stream
(GEMM + its split-K epilogue kernel).  `auto` = the launcher's own choice (splitk_slices)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rassengine_amd import _native as N_

L = N_.lib()
SHAPES = [("qkv", 3072, 1024, 0), ("attn-out", 1024, 1024, 1), ("ffn-up", 4096, 1024, 2), ("ffn-down", 1024, 4096, 1)]
ROWS = [int(v) for v in sys.argv[1:]] or [128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 3072]
ws = torch.empty((64 << 20,), dtype=torch.float32, device="cuda")
stream = ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))


def run(M, N, K, epi, iters=200):
    M_pad = (M + 255) // 256 * 256
    X = torch.randn((M_pad, K), device="cuda").bfloat16()
    W = (torch.randn((N, K), device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn((N,), device="cuda") * 0.1
    R = torch.randn((M_pad, N), device="cuda").bfloat16()
    Y = torch.empty((M_pad, N), dtype=torch.bfloat16, device="cuda")

    def go():
        N_.check("g", L.rass_gemm_bf16_ws(ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
                                        ctypes.c_void_p(R.data_ptr()) if epi == 1 else None, ctypes.c_void_p(Y.data_ptr()), M, M_pad,
                                        N, K, epi, ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4, stream))
    for _ in range(10):
        go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        go()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for M in ROWS:
    for name, N, K, epi in SHAPES:
        os.environ.pop("RASS_GEMM_SPLITK_S", None)
        auto = run(M, N, K, epi)
        res = []
        for S in (1, 2, 4, 8, 16):
            if (K // 64) % S or K // 64 // S < 1:
                continue
            os.environ["RASS_GEMM_SPLITK_S"] = str(S)      # 1 = not split (whole K in one workgroup, fused epilogue)
            res.append(f"S{S} {run(M, N, K, epi):6.1f}")
        os.environ.pop("RASS_GEMM_SPLITK_S", None)
        print(f"M {M:5d} {name:9s} auto {auto:6.1f} us | forced: " + "  ".join(res), flush=True)
