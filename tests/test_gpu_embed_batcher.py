"""The embed micro-batcher on the HIP encoder: N concurrent ``embed_query`` coroutines on one event
loop — what N simultaneous ``/ask`` requests are at app/main.py:2800 — must become one or two encoder forwards
(``rass_encoder_stats``), every caller must get ITS vector, and the whole burst must cost far less than N serial
forwards.

Tolerance (written here): a forward of 32 sequences runs other GEMM / attention kernels than a forward of one (few-rows
kernels vs split-K tiles), with other fp32 summation orders and other points where activations are rounded
to bf16; both results are within cosine 0.999 of the fp32 oracle (tests/test_gpu_cfg3.py, test_gpu_encoder.py), and on
12-token queries of the seeded random model they agree with each other to 0.9998 (measured), so the bar is
COS_SAME_TEXT = 0.9995 — half the distance either is allowed from the oracle."""
import asyncio
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COS_SAME_TEXT = 0.9995

WORDS = ("patient history of diabetes blood pressure note about heart condition drug pain type what is the with for "
         "in on topic number chunk").split()


def _cos(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return np.sum(a * b, axis=-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


@pytest.fixture()
def hip_embedder(large_model):
    from rassengine_amd import embedding
    _, enc = large_model
    embedding.reset_batcher()
    embedding.set_embedder(enc)
    yield enc
    embedding.reset_batcher()
    embedding.set_embedder(None)


def test_32_concurrent_embed_queries_become_one_or_two_forwards(hip_embedder):
    from rassengine_amd import embedding
    enc = hip_embedder
    rng = np.random.default_rng(5)
    queries = [" ".join(rng.choice(WORDS, size=int(rng.integers(6, 14)))) for _ in range(32)]

    async def burst(qs):
        return await asyncio.gather(*[embedding.embed_query(q) for q in qs])

    asyncio.run(burst(queries))                                   # warm: worker thread, workspace, kernels
    lone = []
    t_single = []
    for q in queries:
        t0 = time.perf_counter()
        lone.append(asyncio.run(embedding.embed_query(q)))
        t_single.append(time.perf_counter() - t0)
    single = float(np.median(t_single))

    # five bursts: the forward count is judged on EVERY burst (<= 4) and on the median (<= 2: the worker wakes at the first
    # submit and normally finds all 32 queued, or runs one short forward first and the rest in a second one); only the
    # wall-time criterion takes the fastest burst (a host hiccup — another process on the box, a page fault — must not fail it)
    trials = []
    for _ in range(5):
        s0 = enc.stats()
        t0 = time.perf_counter()
        got_t = asyncio.run(burst(queries))
        wall_t = time.perf_counter() - t0
        s1 = enc.stats()
        assert s1["sequences"] - s0["sequences"] == 32
        trials.append((wall_t, s1["forwards"] - s0["forwards"], got_t))
    assert all(1 <= f <= 4 for _, f, _ in trials), [f for _, f, _ in trials]
    assert float(np.median([f for _, f, _ in trials])) <= 2, [f for _, f, _ in trials]
    wall, forwards, got = min(trials, key=lambda t: t[0])
    worst = 1.0
    for g, l in zip(got, lone):
        assert g.shape == (1, 1024) and g.dtype == np.float32 and np.all(np.isfinite(g))
        worst = min(worst, float(_cos(g, l)[0]))
    print(f"32 concurrent embed_query: {forwards} forward(s), {wall * 1e3:.2f} ms for the burst vs {single * 1e3:.2f} ms "
          f"for one lone query; worst cosine to the lone result {worst:.7f}")
    assert worst >= COS_SAME_TEXT
    assert wall < 4 * single, (wall, single)

    # the vectors are the callers' own: distinct queries -> distinct vectors, in the callers' order
    g = np.concatenate(got)
    sim = _cos(g[:, None, :], np.concatenate(lone)[None, :, :])
    assert np.array_equal(np.argmax(sim, axis=1), np.arange(32))


def test_mixed_requests_on_the_hip_encoder(hip_embedder):
    """Queries, single texts and a short upload in flight together: order, blank rows and shapes of the reference's
    three functions (app/main.py:225-274) on the real encoder."""
    from rassengine_amd import embedding
    enc = hip_embedder
    texts = ["blood pressure note", " ", "history of diabetes with heart condition", "", "pain drug type"]

    async def go():
        return await asyncio.gather(embedding.embed_texts_in_batches(texts), embedding.ollama_embed_text(texts[0]),
                                    embedding.embed_query(texts[2]), embedding.embed_query("   "))

    s0 = enc.stats()
    e, t, q, blank = asyncio.run(go())
    s1 = enc.stats()
    assert s1["forwards"] - s0["forwards"] <= 2 and s1["sequences"] - s0["sequences"] == 5
    assert e.shape == (5, 1024) and not e[1].any() and not e[3].any() and blank.size == 0
    assert _cos(e[0], np.float32(t)) >= COS_SAME_TEXT and _cos(e[2], q[0]) >= COS_SAME_TEXT
    direct = enc.encode([texts[0], texts[2], texts[4]])
    assert np.all(_cos(e[[0, 2, 4]], direct) >= COS_SAME_TEXT)
