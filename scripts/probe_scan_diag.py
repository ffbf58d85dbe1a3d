"""Diagnostic build only (librass_hip_diag.so, built from a patched copy of scan_topk.hip): per-wave counters
of the top-k insertion inside the B = 32 scan."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rassengine_amd.engine import Engine
from rassengine_amd import _native as N
n, dim, k = 1_000_000, 1024, 10
eng = Engine(0, dim); idx = eng.open_index("probe", n); idx.fill_synthetic(n, 1234); eng.synchronize()
for b in (32, 16):
    q = torch.randn((b, dim), device="cuda")
    os_ = torch.empty((b, k), device="cuda"); oi = torch.empty((b, k), dtype=torch.int64, device="cuda")
    for _ in range(20):
        idx.search_device(q.data_ptr(), b, k, os_.data_ptr(), oi.data_ptr())
    eng.synchronize()
    buf = np.zeros(256 * 8 * 4, dtype=np.uint64)
    L = ctypes.CDLL(os.environ["RASS_HIP_LIB"])
    rc = L.rassdiag_read(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    d = buf.reshape(256, 8, 4).astype(np.float64)
    it, ent, cyc, tot = d[..., 0], d[..., 1], d[..., 2], d[..., 3]
    print(f"B={b}: rc={rc} iterations/wave mean {it.mean():.1f} max {it.max():.0f}; loop entries/wave {ent.mean():.1f}; "
          f"s_memtime ticks in insert/wave {cyc.mean():.0f} ({100 * cyc.mean() / tot.mean():.2f}% of the wave's {tot.mean():.0f}); "
          f"ticks per iteration {cyc.sum() / max(it.sum(), 1):.1f}; per-WG max-wave share {100 * (cyc.max(axis=1) / tot.max(axis=1)).mean():.2f}%")
