"""Does a hipGraph of one forward beat the eager launch sequence for a one-query batch?  (experiment)"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd import _native as N
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, weight_names

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--seqlen", type=int, default=16)
ap.add_argument("--iters", type=int, default=300)
a = ap.parse_args()
cfg = EncoderConfig(pooling="mean")
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
lens = np.full(a.batch, a.seqlen)
ids = rng.integers(0, cfg.vocab_size, size=int(lens.sum())).astype(np.int32)
cu = np.zeros(a.batch + 1, dtype=np.int32); np.cumsum(lens, out=cu[1:])
d_ids = torch.from_numpy(ids).cuda(); d_cu = torch.from_numpy(cu).cuda()
out = torch.empty((a.batch, H), device="cuda")
L = N.lib()
def run(stream):
    N.check("enc", L.rass_encode_device(enc._h, ctypes.c_void_p(d_ids.data_ptr()), ctypes.c_void_p(d_cu.data_ptr()),
            a.batch, int(lens.sum()), int(lens.max()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    st = int(s.cuda_stream)
    for _ in range(3): run(st)
    s.synchronize()
    ref = out.clone()
    t0 = time.perf_counter()
    for _ in range(a.iters): run(st)
    s.synchronize()
    eager = (time.perf_counter() - t0) / a.iters
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        run(st)
    g.replay(); s.synchronize()
    same = bool(torch.equal(out, ref))
    t0 = time.perf_counter()
    for _ in range(a.iters): g.replay()
    s.synchronize()
    graph = (time.perf_counter() - t0) / a.iters
print(f"batch={a.batch} seqlen={a.seqlen}: eager {eager*1e3:.3f} ms  graph replay {graph*1e3:.3f} ms  same bits {same}", flush=True)
