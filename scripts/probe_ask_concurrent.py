"""Latency of /ask-shaped requests under CONCURRENT users, through the drop-in boundary (one GPU).

Every simulated user is a coroutine on ONE event loop (uvicorn's model) that repeats what ``ask()`` does on this
path (app/main.py:2800-2885): ``query_emb = await embed_query(text)`` (the embed micro-batcher coalesces here),
``await ensure_index_exists(client, index)`` (the k-NN prefetch shares scan launches here, rassengine_amd/prefetch.py),
then the SYNCHRONOUS ``OpenSearchIndexer(client, index).semantic_search(query_emb=, k=, query=)`` inline on the loop.
Reported per user count: p50 / p99 of one request, requests/s, how many encoder forwards the embeds became
(``rass_encoder_stats``) and how many scan launches the searches became (``rass_engine_kernel_timing_*``), with both
coalescers on, with the k-NN prefetch off (``RASS_KNN_PREFETCH=0``: round 3) and with both off (round 2).
BERT-large-class seeded random weights (no real weights offline), real C++ tokeniser on synthetic text."""
import argparse
import asyncio
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rassengine_amd import config, embedding, indexer, prefetch
from rassengine_amd.docstore import REGISTRY
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
from rassengine_amd.engine import Engine

WORDS = ("patient history of diabetes blood pressure note about heart condition drug pain type what is the with for "
         "in on topic number chunk").split()


def make_queries(n, rng, words=10):
    return [" ".join(rng.choice(WORDS, size=words)) for _ in range(n)]


async def user(name, queries, k, lat):
    for q in queries:
        t0 = time.perf_counter()
        emb = await embedding.embed_query(q)                      # app/main.py:2800
        await indexer.ensure_index_exists(None, name)              # 2801: the k-NN prefetch shares scans here
        hits = indexer.HipIndexer(None, name).semantic_search(query_emb=emb, k=k, query=q)     # 2802, 2878-2885: sync
        lat.append(time.perf_counter() - t0)
        assert len(hits) == k


async def run_users(name, n_users, per_user, k, rng):
    lat = []
    qs = [make_queries(per_user, rng) for _ in range(n_users)]
    t0 = time.perf_counter()
    await asyncio.gather(*[user(name, q, k, lat) for q in qs])
    return np.array(lat) * 1e3, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[10_000, 1_000_000])
    ap.add_argument("--users", type=int, nargs="+", default=[1, 8, 32])
    ap.add_argument("--requests", type=int, default=1500, help="total requests per measurement")
    ap.add_argument("--k", type=int, default=5)
    args = ap.parse_args()

    d = tempfile.mkdtemp(prefix="rass_conc_")
    write_random_model_dir(d, EncoderConfig(pooling="mean"), seed=3)
    enc = HipSentenceEncoder.from_dir(d, device=0)
    embedding.set_embedder(enc)
    eng = Engine.get(config.RASS_DEVICE, 1024)
    rng = np.random.default_rng(0)
    ntok = np.diff(enc.tokenize(make_queries(200, rng))[1])
    print(f"queries: 10 words = {ntok.mean():.1f} tokens on average (min {ntok.min()}, max {ntok.max()})", flush=True)
    for n in args.rows:
        name = f"conc{n}"
        st = REGISTRY.get(name, create=True)
        st.index.fill_synthetic(n, seed=7)
        st.row_doc = [{"doc_id": f"doc-{i}"} for i in range(n)]
        st.doc_row = {f"doc-{i}": i for i in range(n)}
        eng.synchronize()
        for label, batch_max, knn in (("embed+knn shared", 64, 1), ("embed shared (r3) ", 64, 0), ("nothing shared (r2)", 0, 0)):
            config.RASS_EMBED_BATCH_MAX = batch_max
            config.RASS_KNN_PREFETCH = knn
            embedding.reset_batcher()
            for u in args.users:
                per_user = max(20, args.requests // u)
                asyncio.run(run_users(name, u, 20, args.k, rng))          # warm-up
                s0 = enc.stats()
                prefetch.reset_stats()
                eng.kernel_timing_begin(u * per_user + 64)
                lat, wall = asyncio.run(run_users(name, u, per_user, args.k, rng))
                _, scans = eng.kernel_timing_end()
                s1 = enc.stats()
                fw = s1["forwards"] - s0["forwards"]
                print(f"rows {n:8d}  {label}  users {u:3d}: p50 {np.percentile(lat, 50):6.3f} ms  "
                      f"p99 {np.percentile(lat, 99):6.3f} ms  {len(lat) / wall:8.0f} requests/s  "
                      f"{len(lat)} embeds in {fw} forwards ({len(lat) / fw:.1f} per forward), searches in {scans} scan "
                      f"launches ({len(lat) / max(scans, 1):.1f} per launch; answered from a shared scan: "
                      f"{prefetch.stats['answered']})", flush=True)
        REGISTRY.drop(name)
        eng.drop_index(name)
    embedding.reset_batcher()
    embedding.set_embedder(None)
    enc.close()


if __name__ == "__main__":
    main()
