"""Forward latency of the HIP sentence encoder over the small / mid-size shapes the embed micro-batcher produces
(N concurrent queries of ~12 tokens = one varlen forward of N x 12 tokens), device-resident, BERT-large class."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd import _native as N
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, random_weights

shapes = [(1, 12), (2, 12), (4, 12), (8, 12), (16, 12), (32, 12), (64, 12), (32, 32), (16, 64), (8, 128), (4, 512), (16, 512)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
cfg = EncoderConfig(pooling="mean")
enc = HipSentenceEncoder(cfg, random_weights(cfg, 1), None, device=0)
L = N.lib()
rng = np.random.default_rng(0)
stream = enc.stream
for nseq, slen in shapes:
    ids = torch.from_numpy(rng.integers(0, cfg.vocab_size, size=nseq * slen).astype(np.int32)).cuda()
    cu = torch.arange(0, (nseq + 1) * slen, slen, dtype=torch.int32, device="cuda")
    out = torch.empty((nseq, cfg.hidden), device="cuda")
    torch.cuda.synchronize()
    def run():
        N.check("enc", L.rass_encode_device(enc._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(cu.data_ptr()), nseq,
                                            nseq * slen, slen, ctypes.c_void_p(out.data_ptr()), None))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    iters = 100 if nseq * slen <= 2048 else 20
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(f"{nseq:3d} x {slen:3d} = {nseq * slen:5d} tokens: {ms:7.3f} ms per forward = {ms / 24 * 1e3:6.1f} us per layer", flush=True)
enc.close()
