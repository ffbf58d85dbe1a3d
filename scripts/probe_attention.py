"""Times the encoder's attention kernel(s) alone at the cfg-3 shape (256 x 512 tokens, 16 heads of 64), variants
interleaved in one process (RASS_ATTN_VARIANT is read per launch)."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd import _native as N

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--seqlen", type=int, default=512)
ap.add_argument("--varlen", action="store_true")
ap.add_argument("--normal", default="", help="MEAN:STD of the sequence lengths (clipped to 16..512)")
ap.add_argument("--heads", type=int, default=16)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--variants", default="w8,w16")
a = ap.parse_args()
rng = np.random.default_rng(0)
lens = rng.integers(64, 513, size=a.batch) if a.varlen else np.full(a.batch, a.seqlen)
if a.normal:
    mu, sd = (float(v) for v in a.normal.split(":"))
    lens = np.clip(rng.normal(mu, sd, size=a.batch).round().astype(np.int64), 16, 512)
    print(f"lengths ~ N({mu}, {sd}) clipped: mean {lens.mean():.0f}, min {lens.min()}, max {lens.max()}", flush=True)
cu = np.zeros(a.batch + 1, dtype=np.int32); np.cumsum(lens, out=cu[1:])
T = int(cu[-1]); H = a.heads * 64
qkv = torch.randn((T, 3 * H), device="cuda").bfloat16()
ctx = torch.empty((T, H), dtype=torch.bfloat16, device="cuda")
d_cu = torch.from_numpy(cu).cuda()
L = N.lib()
st = ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))
def run():
    N.check("attn", L.rass_attention_bf16(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(d_cu.data_ptr()), a.batch, T,
                                          int(lens.max()), H, a.heads, ctypes.c_void_p(ctx.data_ptr()), st))
flops = 4.0 * 64 * a.heads * float((lens.astype(np.float64) ** 2).sum())
for rnd in range(a.rounds):
    for v in a.variants.split(","):
        if v: os.environ["RASS_ATTN_VARIANT"] = v
        else: os.environ.pop("RASS_ATTN_VARIANT", None)
        run(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print(f"round {rnd} variant={v or 'default':8s} {us:8.1f} us/launch  {flops/us/1e6:7.1f} TFLOP/s", flush=True)
