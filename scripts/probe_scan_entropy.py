"""Is the B = 32 flat scan limited by the chip's clock under load (DVFS give-back, MI355X_MICROARCH.md) or by
its own structure?  Same kernel, same bytes, same launch — only the DATA differs: N(0,1) unit rows (the bench
corpus), a constant corpus (1/sqrt(dim) everywhere: same arithmetic, almost no operand toggling) and zeros.
If the constant / zero corpus runs at the B <= 16 (HBM-bound) time, the B = 32 gap is the clock the chip holds
with the fp32 MFMA pipe ~70 % busy on random operands while HBM streams, not a scheduling problem."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rassengine_amd import ops
from rassengine_amd.engine import HipTimer

n, dim, k, iters = 1_000_000, 1024, 10, 100
torch.manual_seed(0)
slabs = {}
x = torch.randn((n, dim), device="cuda")
x /= x.norm(dim=1, keepdim=True)
slabs["random N(0,1) unit rows"] = ops.pack_rows(x)
del x
slabs["constant 1/32"] = torch.full((n, dim), 1.0 / 32.0, device="cuda")
slabs["zeros"] = torch.zeros((n, dim), device="cuda")
stream = int(torch.cuda.current_stream().cuda_stream)
for name, slab in slabs.items():
    for qname, qgen in (("random queries", lambda b: torch.randn((b, dim), device="cuda")),
                        ("zero queries", lambda b: torch.zeros((b, dim), device="cuda"))):
        row = []
        for b in (32, 16):
            q = qgen(b)
            for _ in range(30):
                ops.scan_topk_packed(slab, n, q, k)
            torch.cuda.synchronize()
            t = HipTimer()
            t.start(stream)
            for _ in range(iters):
                ops.scan_topk_packed(slab, n, q, k)
            t.stop(stream)
            us = t.elapsed_ms() / iters * 1e3
            row.append(f"B={b}: {us:7.1f} us/step ({n * dim * 4 / us / 1e6:6.0f} GB/s incl. merge)")
        print(f"{name:28s} {qname:15s} " + "   ".join(row), flush=True)
