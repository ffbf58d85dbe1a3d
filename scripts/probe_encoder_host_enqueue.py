"""Host side of one encoder forward: how long rass_encode_device takes to ENQUEUE its launches (into an empty queue) against the
forward's GPU time — is the one-query forward bound by the host's launch rate?"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rassengine_amd import _native as N
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, random_weights
cfg = EncoderConfig(pooling="mean")
enc = HipSentenceEncoder(cfg, random_weights(cfg, 1), None, device=0)
L = N.lib()
rng = np.random.default_rng(0)
for nseq, slen in [(1, 12), (32, 12)]:
    ids = torch.from_numpy(rng.integers(0, cfg.vocab_size, size=nseq * slen).astype(np.int32)).cuda()
    cu = torch.arange(0, (nseq + 1) * slen, slen, dtype=torch.int32, device="cuda")
    out = torch.empty((nseq, cfg.hidden), device="cuda")
    def run():
        N.check("enc", L.rass_encode_device(enc._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(cu.data_ptr()), nseq, nseq * slen, slen, ctypes.c_void_p(out.data_ptr()), None))
    for _ in range(10): run()
    torch.cuda.synchronize()
    iters = 200
    t0 = time.perf_counter()
    for _ in range(iters): run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{nseq} x {slen}: host enqueue {1e3*(t1-t0)/iters:.3f} ms per forward, total {1e3*(t2-t0)/iters:.3f} ms per forward", flush=True)
    # one forward at a time (the lone-user case): enqueue + sync
    lat, host = [], []
    for _ in range(200):
        t0 = time.perf_counter(); run(); t1 = time.perf_counter(); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0); host.append(t1 - t0)
    print(f"   one at a time: p50 {1e3*np.median(lat):.3f} ms, of which the host call (enqueue into an empty queue) {1e3*np.median(host):.3f} ms")
enc.close()
