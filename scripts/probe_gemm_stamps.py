"""Where a tile of the persistent GEMM (gemm_bf16_p5_kernel) spends its time: wall-clock stamps of workgroups 0..7, wave 0
(library built with -DRASS_GEMM_STAMPS: make EXTRA=-DRASS_GEMM_STAMPS into another file, RASS_HIP_LIB=...).  Per tile:
K loop, epilogue until its last store is ISSUED, until the stores have LEFT (a diagnostic vmcnt(0)), gap to the next tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd import _native as N
L = N.lib()
g = torch.Generator(device="cuda"); g.manual_seed(0)
M = 131072
stream = ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))
for (Nn, K, epi, name) in [(3072, 1024, 0, "QKV"), (1024, 1024, 1, "attn-out"), (4096, 1024, 2, "FFN-up"), (1024, 4096, 1, "FFN-down")]:
    X = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((Nn, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn((Nn,), generator=g, device="cuda")
    R = torch.randn((M, Nn), generator=g, device="cuda").bfloat16()
    Y = torch.empty((M, Nn), dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        N.check("g", L.rass_gemm_bf16(ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                ctypes.c_void_p(R.data_ptr()), ctypes.c_void_p(Y.data_ptr()), M, M, Nn, K, epi, stream))
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * (8 * 64 * 4))()
    rc = ctypes.CDLL(N.LIB_PATH).rassdiag_gemm_stamps(out, 8 * 64 * 4)
    assert rc == 0, rc
    st = np.array(list(out), dtype=np.float64).reshape(8, 64, 4) * 0.01     # us (100 MHz)
    tiles = (Nn // 256) * (M // 256) // 256
    for blk in (0, 3):
        s = st[blk, :tiles]
        kloop = s[:, 1] - s[:, 0]
        epi_issue = s[:, 2] - s[:, 1]
        drain = s[:, 3] - s[:, 2]
        gap = s[1:, 0] - s[:-1, 3]
        print(f"{name:9s} N={Nn} K={K} block {blk}: {tiles} tiles; K loop {kloop[1:-1].mean():6.2f} us ({K // 64} steps: {kloop[1:-1].mean() / (K // 64):.3f} per step), "
              f"epilogue to last store issued {epi_issue[1:-1].mean():6.2f}, stores drained +{drain[1:-1].mean():5.2f}, gap to next tile {gap[1:].mean():5.2f}; "
              f"whole launch {s[-1, 3] - s[0, 0]:7.1f} us", flush=True)
