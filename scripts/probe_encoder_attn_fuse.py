"""Query-time forward with the attention recomputed inside the attention-output GEMM (RASS_ATTN_FUSE, default on) against
the attention launch + GEMM pair (RASS_ATTN_FUSE=0): latency per forward and agreement of the pooled embeddings, same
process, same device (the variable is read per launch).  BERT-large class, seeded random weights."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd import _native as N
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, random_weights

cases = [[12], [16], [8], [23], [32], [12, 12], [9, 7], [12, 15, 5], [8, 8, 8, 8]]
cfg = EncoderConfig(pooling="mean")
enc = HipSentenceEncoder(cfg, random_weights(cfg, 1), None, device=0)
L = N.lib()
rng = np.random.default_rng(0)


def forward_ms(ids, cu, nseq, total, max_len, out, iters=200):
    def run():
        N.check("enc", L.rass_encode_device(enc._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(cu.data_ptr()), nseq, total,
                                            max_len, ctypes.c_void_p(out.data_ptr()), None))
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    return best


for lens in cases:
    total = sum(lens)
    ids = torch.from_numpy(rng.integers(0, cfg.vocab_size, size=total).astype(np.int32)).cuda()
    cu = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    res = {}
    for fuse in ("0", "2", "0", "2"):
        os.environ["RASS_ATTN_FUSE"] = fuse
        out = torch.empty((len(lens), cfg.hidden), device="cuda")
        ms = forward_ms(ids, cu, len(lens), total, max(lens), out)
        res.setdefault(fuse, []).append((ms, out.cpu().numpy()))
    a, b = res["0"][0][1], res["2"][0][1]
    cos = (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)
    print(f"lens {str(lens):16s} pair {min(m for m, _ in res['0']):.4f} ms   fused {min(m for m, _ in res['2']):.4f} ms   "
          f"min cos(pair, fused) {cos.min():.7f}   max |diff| {np.abs(a - b).max():.2e}   "
          f"deterministic {np.array_equal(res['2'][0][1], res['2'][1][1])}", flush=True)
os.environ.pop("RASS_ATTN_FUSE", None)
enc.close()
