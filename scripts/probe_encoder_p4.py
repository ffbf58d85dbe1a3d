"""The whole encoder forward (256 x 512 tokens, LN fold on) with the persistent GEMMs as p5 and as p4 (RASS_GEMM_VARIANT): the
pooled outputs must be bit-identical (same MFMA order, same epilogue arithmetic); ms per forward of each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, random_weights
cfg = EncoderConfig(pooling="mean")
enc = HipSentenceEncoder(cfg, random_weights(cfg, seed=1), None, device=0)
nseq, S = int(os.environ.get("P4_NSEQ", 256)), 512
g = torch.Generator(device="cuda"); g.manual_seed(1)
ids = torch.randint(1000, cfg.vocab_size - 1, (nseq * S,), generator=g, device="cuda", dtype=torch.int32)
cu = torch.arange(0, (nseq + 1) * S, S, dtype=torch.int32, device="cuda")
out = {}
for variant in ("p5", "p4", "default", "p5", "p4", "default"):
    if variant == "default":
        os.environ.pop("RASS_GEMM_VARIANT", None)
    else:
        os.environ["RASS_GEMM_VARIANT"] = variant
    o = torch.empty((nseq, cfg.hidden), dtype=torch.float32, device="cuda")
    for _ in range(2):
        enc.encode_device(ids.data_ptr(), cu.data_ptr(), nseq, nseq * S, S, o.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        enc.encode_device(ids.data_ptr(), cu.data_ptr(), nseq, nseq * S, S, o.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    out[variant] = o.cpu().numpy()
    print(f"{variant}: {ms:.2f} ms per forward = {nseq / ms * 1e3:.0f} chunks/s", flush=True)
same = np.array_equal(out["p5"].view(np.uint32), out["p4"].view(np.uint32))
print("bit-identical pooled outputs:", same, "max |diff|", float(np.abs(out["p5"] - out["p4"]).max()), "finite", bool(np.isfinite(out["p4"]).all()))
