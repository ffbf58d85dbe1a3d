"""What one /ask spends in this engine: embed_query (host ids in, host vector out: rass_encode) + semantic_search over an
index (host query in, host top-k out: rass_index_search_ex), serial, one GPU.  BERT-large-class random weights."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
from rassengine_amd.engine import Engine

d = tempfile.mkdtemp(prefix="rass_lat_")
write_random_model_dir(d, EncoderConfig(pooling="mean"), seed=3)
enc = HipSentenceEncoder.from_dir(d, device=0)
eng = Engine(0, 1024)
rng = np.random.default_rng(0)
for n in (10_000, 1_000_000):
    idx = eng.open_index(f"ask{n}", capacity_rows=n)
    idx.fill_synthetic(n, seed=7)
    eng.synchronize()
    for toks in (12, 20, 48):
        seqs = [[101] + list(rng.integers(1000, 30000, size=toks - 2)) + [102] for _ in range(300)]
        for s in seqs[:20]:
            idx.search(enc.encode_ids([s]), 5)
        te, ts = [], []
        for s in seqs:
            t0 = time.perf_counter(); v = enc.encode_ids([s]); t1 = time.perf_counter(); idx.search(v, 5); t2 = time.perf_counter()
            te.append(t1 - t0); ts.append(t2 - t1)
        te, ts = np.array(te) * 1e3, np.array(ts) * 1e3
        print(f"rows {n:8d}  query of {toks:2d} tokens: embed p50 {np.percentile(te,50):.3f} ms (p99 {np.percentile(te,99):.3f})  "
              f"search k=5 p50 {np.percentile(ts,50):.3f} ms (p99 {np.percentile(ts,99):.3f})  total p50 {np.percentile(te+ts,50):.3f} ms",
              flush=True)
    eng.drop_index(f"ask{n}")
enc.close(); eng.close()
