"""Perf probe of the HIP sentence encoder (BERT-large class, random weights)."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd import _native as N
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, weight_names

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--seqlen", type=int, default=512)
ap.add_argument("--varlen", action="store_true")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--layers", type=int, default=24)
a = ap.parse_args()

cfg = EncoderConfig(layers=a.layers, pooling="mean")
rng = np.random.default_rng(0)
H, I = cfg.hidden, cfg.intermediate
w = {}
for name in weight_names(cfg.layers):
    if name.endswith("word_embeddings.weight"): shape = (cfg.vocab_size, H)
    elif name.endswith("position_embeddings.weight"): shape = (cfg.max_positions, H)
    elif name.endswith("token_type_embeddings.weight"): shape = (2, H)
    elif "LayerNorm" in name: shape = (H,)
    elif name.endswith(".bias"): shape = (I,) if "intermediate" in name else (H,)
    elif "intermediate.dense" in name: shape = (I, H)
    elif ".output.dense" in name and "attention" not in name: shape = (H, I)
    else: shape = (H, H)
    t = rng.standard_normal(shape, dtype=np.float32) * (0.03 if len(shape) == 2 else 0.05)
    if "LayerNorm.weight" in name: t = 1 + t
    w[name] = t
enc = HipSentenceEncoder(cfg, w, None, device=0)
lens = rng.integers(64, 513, size=a.batch) if a.varlen else np.full(a.batch, a.seqlen)
ids = rng.integers(0, cfg.vocab_size, size=int(lens.sum())).astype(np.int32)  # random tokens (not zeros)
cu = np.zeros(a.batch + 1, dtype=np.int32); np.cumsum(lens, out=cu[1:])
d_ids = torch.from_numpy(ids).cuda(); d_cu = torch.from_numpy(cu).cuda()
out = torch.empty((a.batch, H), device="cuda")
L = N.lib()
stream = int(torch.cuda.current_stream().cuda_stream)
def run():
    N.check("enc", L.rass_encode_device(enc._h, ctypes.c_void_p(d_ids.data_ptr()), ctypes.c_void_p(d_cu.data_ptr()),
            a.batch, int(lens.sum()), int(lens.max()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)))
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
T = int(lens.sum())
flops = cfg.layers * (2 * T * (4 * H * H + 2 * H * I)) + cfg.layers * 4 * H * float((lens.astype(np.float64) ** 2).sum())
print(f"batch={a.batch} tokens={T} {dt*1e3:.1f} ms/forward  {a.batch/dt:.0f} chunks/s  {T/dt:.0f} tok/s  "
      f"{flops/dt/1e12:.1f} TFLOP/s = {100*flops/dt/2.5e15:.1f}% of 2.5 PF", flush=True)
