"""Calibration only (the library is NOT on the product path): the encoder's attention shape — 256 sequences x 512 tokens, 16 heads
of 64, bf16, no mask — through this repo's attention64_kernel and through torch.nn.functional.scaled_dot_product_attention
(AOTriton / CK flash attention on ROCm), same device, same process."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from rassengine_amd import _native as N

B, S, Hh, D = 256, 512, 16, 64
H = Hh * D
T = B * S
qkv = torch.randn((T, 3 * H), device="cuda").bfloat16()
ctx = torch.empty((T, H), dtype=torch.bfloat16, device="cuda")
cu = torch.from_numpy((np.arange(B + 1) * S).astype(np.int32)).cuda()
L = N.lib()
st = ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))
def ours():
    N.check("attn", L.rass_attention_bf16(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(cu.data_ptr()), B, T, S, H, Hh,
                                          ctypes.c_void_p(ctx.data_ptr()), st))
q = qkv[:, :H].reshape(B, S, Hh, D).transpose(1, 2).contiguous()
k = qkv[:, H:2 * H].reshape(B, S, Hh, D).transpose(1, 2).contiguous()
v = qkv[:, 2 * H:].reshape(B, S, Hh, D).transpose(1, 2).contiguous()
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
flops = 4.0 * S * S * D * Hh * B
from torch.nn.attention import SDPBackend, sdpa_kernel
res = {}
for rnd in range(2):
    t = timed(ours); print(f"round {rnd}: attention64_kernel            {t:7.1f} us  {flops / t / 1e6:6.0f} TFLOP/s", flush=True)
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION)):
        try:
            with sdpa_kernel(be):
                t = timed(lambda: F.scaled_dot_product_attention(q, k, v))
            print(f"round {rnd}: torch SDPA {name:10s}           {t:7.1f} us  {flops / t / 1e6:6.0f} TFLOP/s", flush=True)
        except Exception as ex:  # noqa: BLE001
            print(f"round {rnd}: torch SDPA {name}: not available here ({type(ex).__name__}: {str(ex)[:120]})", flush=True)
# agreement (sanity): our output against SDPA's on the same inputs
ours(); torch.cuda.synchronize()
ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(T, H)
print("max |ours - sdpa| =", float((ctx.float() - ref.float()).abs().max()))
