"""Repeat small encodes (the split-K query-time path) and check that the same text always gives the same bits, alone
and inside batches of changing composition."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
hidden = int(os.environ.get("H", 128))
with tempfile.TemporaryDirectory() as d:
    write_random_model_dir(d, EncoderConfig(vocab_size=300, hidden=hidden, layers=2 if hidden <= 256 else 4, heads=hidden // 64,
                                            intermediate=4 * hidden, max_positions=512), seed=11)
    enc = HipSentenceEncoder.from_dir(d, device=0)
    texts = ["patient history of diabetes", "blood pressure note", "chunk number 3 about topic", "a", "heart pain with type 2 diabetes history"]
    ref = {t: enc.encode([t])[0].copy() for t in texts}
    rng = np.random.default_rng(0)
    bad = 0
    for it in range(int(os.environ.get("ITERS", 400))):
        k = int(rng.integers(1, 6))
        pick = [texts[i] for i in rng.integers(0, len(texts), size=k)]
        out = enc.encode(pick)
        for t, v in zip(pick, out):
            if not np.array_equal(v, ref[t]):
                bad += 1
                if bad <= 5:
                    print("MISMATCH it", it, "batch", len(pick), "max|d|", float(np.abs(v - ref[t]).max()), flush=True)
    print("hidden", hidden, "mismatches", bad, flush=True)
    enc.close()
