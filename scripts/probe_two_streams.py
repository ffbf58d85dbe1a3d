"""Does the ingest forward gain from two half-batches in flight on two streams (the low-power LayerNorm / attention phases of one
beside the power-limited GEMMs of the other)?  Two encoder instances (own stream and workspace each, same seeded weights),
128 x 512 tokens per forward, enqueued alternately, against one instance at 256 x 512.  Chunks/s of both, same process."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd import _native as N
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, random_weights

cfg = EncoderConfig(layers=24, pooling="mean")
w = random_weights(cfg, seed=0)
encs = [HipSentenceEncoder(cfg, w, None, device=0) for _ in range(2)]
L = N.lib()
rng = np.random.default_rng(1)
H = cfg.hidden

def batch(nb, S=512):
    ids = torch.from_numpy(rng.integers(0, cfg.vocab_size, size=nb * S).astype(np.int32)).cuda()
    cu = torch.from_numpy((np.arange(nb + 1) * S).astype(np.int32)).cuda()
    return ids, cu, nb, nb * S, S

def fwd(enc, b, out):
    ids, cu, nb, total, mx = b
    N.check("enc", L.rass_encode_device(enc._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(cu.data_ptr()), nb, total, mx,
                                        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(enc.stream)))

big = batch(256); halves = [batch(128), batch(128)]
out_big = torch.empty((256, H), device="cuda"); outs = [torch.empty((128, H), device="cuda") for _ in range(2)]
for rnd in range(3):
    for _ in range(2): fwd(encs[0], big, out_big)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): fwd(encs[0], big, out_big)
    torch.cuda.synchronize(); one = (time.perf_counter() - t0) / 6
    for _ in range(2):
        for i in range(2): fwd(encs[i], halves[i], outs[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        for i in range(2): fwd(encs[i], halves[i], outs[i])
    torch.cuda.synchronize(); two = (time.perf_counter() - t0) / 6
    for _ in range(2): fwd(encs[0], halves[0], outs[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(12): fwd(encs[0], halves[0], outs[0])
    torch.cuda.synchronize(); half_serial = (time.perf_counter() - t0) / 6
    print(f"round {rnd}: one stream 256 x 512: {one*1e3:.1f} ms = {256/one:.0f} chunks/s | two streams 2 x (128 x 512): {two*1e3:.1f} ms = {256/two:.0f} chunks/s"
          f" | one stream, 128 x 512 twice: {half_serial*1e3:.1f} ms = {256/half_serial:.0f} chunks/s", flush=True)
