"""Diagnostic build only (a librass_hip built with -DRASS_SCAN_DIAG, named by RASS_HIP_LIB): launch-wide counters of
the top-k ranking inside the flat scan — ranking calls, calls that insert, candidates, shared-floor posts and the
candidates the shared floor rejected."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rassengine_amd.engine import Engine
n, dim, k = int(os.environ.get("N", 1_000_000)), 1024, 10
eng = Engine(0, dim); idx = eng.open_index("probe", n); idx.fill_synthetic(n, 1234); eng.synchronize()
L = ctypes.CDLL(os.environ["RASS_HIP_LIB"])
buf = np.zeros(8, dtype=np.uint64)
for b in (32, 16):
    q = torch.randn((b, dim), device="cuda")
    os_ = torch.empty((b, k), device="cuda"); oi = torch.empty((b, k), dtype=torch.int64, device="cuda")
    reps = 20
    L.rassdiag_read(buf.ctypes.data_as(ctypes.c_void_p), 8)
    for _ in range(reps):
        idx.search_device(q.data_ptr(), b, k, os_.data_ptr(), oi.data_ptr())
    eng.synchronize()
    rc = L.rassdiag_read(buf.ctypes.data_as(ctypes.c_void_p), 8)
    c = buf.astype(np.float64) / reps
    print(f"{os.path.basename(os.environ['RASS_HIP_LIB'])} sample_floor={os.environ.get('RASS_SCAN_SAMPLE_FLOOR', '1')} B={b}: rc={rc} "
          f"per launch: ranking calls {c[0]:.0f}, with candidates {c[1]:.0f} ({100 * c[1] / max(c[0], 1):.1f}%), "
          f"candidates {c[2]:.0f}, floor-rejected candidates {c[4]:.0f}", flush=True)
