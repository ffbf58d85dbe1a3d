"""GPU parity of the sentence-encoder kernels (K4-K8) through the C ABI against the fp32 CPU
oracle (oracle/bert_ref.py: plain PyTorch fp32 forward of the same architecture and weights).

Tolerance (north_star / SURVEY §8c O3): the bf16 GPU path must reach cosine >= 0.999 to the
fp32 oracle on the pooled sentence vector; the GEMM alone is held to bf16 rounding of an fp32
reference (|err| <= 1.5 * 2^-8 * |ref| + small abs)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cos(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return np.sum(a * b, axis=-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


@pytest.mark.parametrize("M,N,K,epi", [(128, 128, 64, 0), (100, 384, 128, 0), (300, 128, 512, 1), (257, 512, 128, 2),
                                       (1000, 3072, 1024, 0), (513, 1024, 4096, 1), (640, 4096, 1024, 2),
                                       # >= 1024 rows, N % 256 == 0, few 256^2 tiles: 128^2 tiles (split over K)
                                       (1024, 256, 64, 0), (1500, 1024, 1024, 1), (4096, 3072, 1024, 0),
                                       (1100, 4096, 1024, 2), (1300, 1024, 4096, 1), (2048, 512, 128, 0),
                                       # >= 192 tiles of 256^2: the persistent kernel (p5) walks several tiles per
                                       # workgroup (next-tile prefetch under the epilogue), ragged last M tile;
                                       # 2 / 3 / 5 / 16 / 64 K steps per tile, odd step counts; K = 64 stays on 128^2 tiles
                                       (70000, 1024, 1024, 1), (33000, 3072, 256, 0), (66000, 512, 64, 2),
                                       (65537, 256, 128, 1), (3000, 512, 128, 1), (70001, 256, 192, 2),
                                       (33333, 768, 320, 1), (1025, 1024, 2048, 0), (40000, 1024, 4096, 1)])
def test_gemm_bf16_matches_torch(gpu, M, N, K, epi):
    torch = gpu
    from rassengine_amd import _native as N_
    g = torch.Generator(device="cuda")
    g.manual_seed(M * 7 + N + K + epi)
    M_pad = (M + 255) // 256 * 256 if M >= 1024 else (M + 127) // 128 * 128
    X = torch.zeros((M_pad, K), dtype=torch.bfloat16, device="cuda")
    X[:M] = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn((N,), generator=g, device="cuda") * 0.1
    R = torch.zeros((M_pad, N), dtype=torch.bfloat16, device="cuda")
    R[:M] = torch.randn((M, N), generator=g, device="cuda").bfloat16()
    Y = torch.full((M_pad, N), 777.0, dtype=torch.bfloat16, device="cuda")
    N_.check("rass_gemm_bf16", N_.lib().rass_gemm_bf16(
        ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
        ctypes.c_void_p(R.data_ptr()) if epi == 1 else None, ctypes.c_void_p(Y.data_ptr()), M, M_pad, N, K, epi,
        ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
    torch.cuda.synchronize()
    ref = X[:M].float() @ W.float().T + bias
    if epi == 1:
        ref = ref + R[:M].float()
    if epi == 2:
        ref = torch.nn.functional.gelu(ref)
    got = Y[:M].float()
    err = (got - ref).abs()
    tol = 1.5 * 2.0 ** -8 * ref.abs() + 2e-3
    assert bool((err <= tol).all()), float((err - tol).max())
    assert bool((Y[M:] == 777.0).all())  # padding rows are never written


_FORCED_VARIANT_SCRIPT = r"""
import ctypes, sys, torch
sys.path.insert(0, %r)
from rassengine_amd import _native as N_
ok = True
for (M, N, K, epi) in [(1024, 256, 128, 0), (3000, 512, 128, 1), (1025, 1024, 2048, 2), (2048, 1024, 4096, 1),
                       (1300, 256, 512, 0), (1024, 256, 1024, 1), (4097, 512, 1024, 2), (1791, 768, 576, 1)]:
    g = torch.Generator(device="cuda"); g.manual_seed(M + N + K + epi)
    M_pad = (M + 255) // 256 * 256
    X = torch.zeros((M_pad, K), dtype=torch.bfloat16, device="cuda"); X[:M] = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn((N,), generator=g, device="cuda") * 0.1
    R = torch.zeros((M_pad, N), dtype=torch.bfloat16, device="cuda"); R[:M] = torch.randn((M, N), generator=g, device="cuda").bfloat16()
    Y = torch.full((M_pad, N), 777.0, dtype=torch.bfloat16, device="cuda")
    N_.check("g", N_.lib().rass_gemm_bf16(ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
             ctypes.c_void_p(R.data_ptr()) if epi == 1 else None, ctypes.c_void_p(Y.data_ptr()), M, M_pad, N, K, epi,
             ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
    torch.cuda.synchronize()
    ref = X[:M].float() @ W.float().T + bias
    if epi == 1: ref = ref + R[:M].float()
    if epi == 2: ref = torch.nn.functional.gelu(ref)
    err = (Y[:M].float() - ref).abs()
    ok = ok and bool((err <= 1.5 * 2.0 ** -8 * ref.abs() + 2e-3).all()) and bool((Y[M:] == 777.0).all())
print("VARIANT_OK" if ok else "VARIANT_BAD")
"""


@pytest.mark.parametrize("variant", ["p5", "p4"])
def test_gemm_forced_variants_on_few_tiles(gpu, variant):
    """By default shapes with few 256^2 tiles take the 128^2 kernel; RASS_GEMM_VARIANT=p5 / p4 (read once per process: hence a
    child process, started before this one's GPU state matters to it) keeps the persistent 256^2 kernel for them — one
    tile per workgroup, fewer tiles than CUs, ragged M, the shortest K p4 takes (512: its two store steps + the stream's
    run-on), K not a multiple of 128 — the 8-wave p5 or the 4-wave p4 (round 4's default for big shapes; K < 512 stays on p5).  (Round 2's other variants — ring, pring, p64, w4l — were retired
    from the library: scripts/microbench/gemm_retired_kernels.hip.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RASS_GEMM_VARIANT=variant)
    out = subprocess.run([sys.executable, "-c", _FORCED_VARIANT_SCRIPT % root], env=env, capture_output=True, text=True,
                         timeout=300)
    assert "VARIANT_OK" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])


@pytest.mark.parametrize("mid_s", ["auto", "2", "4", "8", "16"])
@pytest.mark.parametrize("M,N,K,epi", [(384, 3072, 1024, 0), (200, 1024, 1024, 1), (384, 4096, 1024, 2), (129, 1024, 4096, 1),
                                       (1000, 1024, 4096, 0), (65, 128, 128, 2), (640, 256, 64, 1), (2047, 3072, 1024, 0)])
def test_gemm_mid_size_split_k(gpu, M, N, K, epi, mid_s, monkeypatch):
    """Mid-size batches (65 .. ~2 000 rows: N concurrent queries coalesced by the embed micro-batcher, small uploads) on
    the 128^2 kernels: the slice count the launcher picks (round 3's rule: ~64 workgroups for K = 1 024, ~256 for K = 4 096)
    and every forced count (RASS_GEMM_SPLITK_S, read per launch), against torch within bf16 rounding; bit-identical
    from run to run (fp32 partials summed in fixed order)."""
    torch = gpu
    if mid_s != "auto":
        monkeypatch.setenv("RASS_GEMM_SPLITK_S", mid_s)
    from rassengine_amd import _native as N_
    g = torch.Generator(device="cuda")
    g.manual_seed(M * 13 + N + K + epi)
    M_pad = (M + 255) // 256 * 256
    X = torch.zeros((M_pad, K), dtype=torch.bfloat16, device="cuda")
    X[:M] = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn((N,), generator=g, device="cuda") * 0.1
    R = torch.zeros((M_pad, N), dtype=torch.bfloat16, device="cuda")
    R[:M] = torch.randn((M, N), generator=g, device="cuda").bfloat16()
    ws = torch.empty((16 * 2048 * 1024,), dtype=torch.float32, device="cuda")
    outs = []
    for _ in range(2):
        Y = torch.full((M_pad, N), 777.0, dtype=torch.bfloat16, device="cuda")
        N_.check("rass_gemm_bf16_ws", N_.lib().rass_gemm_bf16_ws(
            ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
            ctypes.c_void_p(R.data_ptr()) if epi == 1 else None, ctypes.c_void_p(Y.data_ptr()), M, M_pad, N, K, epi,
            ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4, ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
        torch.cuda.synchronize()
        outs.append(Y)
    ref = X[:M].float() @ W.float().T + bias
    if epi == 1:
        ref = ref + R[:M].float()
    if epi == 2:
        ref = torch.nn.functional.gelu(ref)
    err = (outs[0][:M].float() - ref).abs()
    tol = 1.5 * 2.0 ** -8 * ref.abs() + 2e-3
    assert bool((err <= tol).all()), float((err - tol).max())
    assert bool((outs[0][M:] == 777.0).all()) and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("fewrows", ["1", "0"])
@pytest.mark.parametrize("M,N,K,epi", [(1, 1024, 1024, 1), (16, 3072, 1024, 0), (37, 4096, 1024, 2), (100, 1024, 4096, 1),
                                       (129, 1024, 4096, 1), (200, 128, 128, 0), (256, 384, 512, 2),
                                       (1300, 1024, 1024, 1), (700, 1024, 4096, 0),
                                       # a few rows against a wide matrix: the one-launch kernel (one wave per 16 features)
                                       (1, 3072, 1024, 0), (64, 2048, 2048, 1), (49, 4096, 1024, 2), (17, 2048, 1024, 1),
                                       (33, 1024, 4096, 1), (64, 1024, 8192, 0), (5, 1024, 3072, 2),
                                       # 65 .. 128 rows: the same kernel with 5 .. 8 row blocks per wave (K <= 3072)
                                       (65, 3072, 1024, 0), (96, 4096, 1024, 2), (100, 2048, 2048, 1), (128, 3072, 1024, 0),
                                       (120, 1024, 3072, 1)])
def test_gemm_split_k_for_few_rows(gpu, M, N, K, epi, fewrows, monkeypatch):
    """The query-time paths (rass_gemm_bf16_ws) against torch, bit-identical from run to run (no atomics): K split over
    workgroups with the slices summed in fixed order, and — N >= 1024 and tokens <= 128 (K <= 3072) or <= 64 (K = 4096, 8192) — the one-launch kernel
    (RASS_GEMM_FEWROWS=0, read per launch, keeps the split-K pair for those shapes too)."""
    torch = gpu
    monkeypatch.setenv("RASS_GEMM_FEWROWS", fewrows)
    from rassengine_amd import _native as N_
    g = torch.Generator(device="cuda")
    g.manual_seed(M * 11 + N + K + epi)
    M_pad = (M + 255) // 256 * 256
    X = torch.zeros((M_pad, K), dtype=torch.bfloat16, device="cuda")
    X[:M] = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn((N,), generator=g, device="cuda") * 0.1
    R = torch.zeros((M_pad, N), dtype=torch.bfloat16, device="cuda")
    R[:M] = torch.randn((M, N), generator=g, device="cuda").bfloat16()
    ws = torch.empty((16 * 256 * 1024,), dtype=torch.float32, device="cuda")
    outs = []
    for _ in range(2):
        Y = torch.full((M_pad, N), 777.0, dtype=torch.bfloat16, device="cuda")
        N_.check("rass_gemm_bf16_ws", N_.lib().rass_gemm_bf16_ws(
            ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
            ctypes.c_void_p(R.data_ptr()) if epi == 1 else None, ctypes.c_void_p(Y.data_ptr()), M, M_pad, N, K, epi,
            ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4, ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))))
        torch.cuda.synchronize()
        outs.append(Y)
    assert torch.equal(outs[0], outs[1])
    ref = X[:M].float() @ W.float().T + bias
    if epi == 1:
        ref = ref + R[:M].float()
    if epi == 2:
        ref = torch.nn.functional.gelu(ref)
    got = outs[0][:M].float()
    err = (got - ref).abs()
    tol = 1.5 * 2.0 ** -8 * ref.abs() + 2e-3
    assert bool((err <= tol).all()), float((err - tol).max())
    assert bool((outs[0][M:] == 777.0).all())  # padding rows are never written


@pytest.fixture(scope="module")
def tiny_model(tmp_path_factory):
    from rassengine_amd.encoder import EncoderConfig, write_random_model_dir
    d = str(tmp_path_factory.mktemp("tiny_model_gpu"))
    write_random_model_dir(d, EncoderConfig(vocab_size=300, hidden=128, layers=2, heads=2, intermediate=512,
                                            max_positions=512), seed=11)
    return d


@pytest.mark.parametrize("pooling", ["cls", "mean"])
def test_tiny_encoder_matches_oracle(gpu, tiny_model, pooling, monkeypatch):
    from oracle import bert_ref
    from rassengine_amd import config
    from rassengine_amd.encoder import HipSentenceEncoder
    monkeypatch.setattr(config, "RASS_POOLING", pooling)
    rng = np.random.default_rng(3)
    lens = [1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 300, 512, 40]
    seqs = [list(rng.integers(0, 300, size=n)) for n in lens]
    enc = HipSentenceEncoder.from_dir(tiny_model, device=0)
    try:
        assert enc.cfg.pooling == pooling
        got = enc.encode_ids(seqs)
        got2 = enc.encode_ids(seqs[::-1])[::-1]  # batching order must not matter
    finally:
        enc.close()
    ref = bert_ref.pool(bert_ref.forward_plain(tiny_model, seqs), pooling)
    ref_bf = bert_ref.pool(bert_ref.forward_plain(tiny_model, seqs, bf16_weights=True), pooling)
    assert got.shape == (len(seqs), 128) and got.dtype == np.float32 and np.all(np.isfinite(got))
    c = _cos(got, ref)
    assert np.all(c >= 0.999), c
    assert np.all(_cos(got, ref_bf) >= 0.999)
    assert np.abs(got - ref).max() <= 0.08 * np.abs(ref).max()
    assert np.array_equal(got, got2)


def test_large_encoder_sample_matches_oracle(gpu, tmp_path_factory):
    """BERT-large-class (24 x 1024 x 16 heads x 4096, vocab 30522): the mxbai-embed-large shape
    with seeded random weights (no real weights offline)."""
    from oracle import bert_ref
    from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
    d = str(tmp_path_factory.mktemp("large_model"))
    cfg = EncoderConfig(pooling="mean")
    write_random_model_dir(d, cfg, seed=1)
    rng = np.random.default_rng(99)
    seqs = [list(rng.integers(0, 30522, size=n)) for n in (32, 20, 7)]
    one = [list(rng.integers(0, 30522, size=11))]      # a query: <= 16 rows, every linear through the few-rows kernels
    two = [list(rng.integers(0, 30522, size=n)) for n in (9, 7)]
    longer = [list(rng.integers(0, 30522, size=23))]          # 17..32 rows: the same kernels on two row blocks
    three = [list(rng.integers(0, 30522, size=n)) for n in (12, 15, 5)]
    enc = HipSentenceEncoder.from_dir(d, device=0)
    try:
        got = enc.encode_ids(seqs)
        got_one = enc.encode_ids(one)
        got_two = enc.encode_ids(two)
        got_longer = enc.encode_ids(longer)
        got_three = enc.encode_ids(three)
    finally:
        enc.close()
    ref = bert_ref.pool(bert_ref.forward_plain(d, seqs), "mean")
    c = _cos(got, ref)
    assert np.all(c >= 0.999), c
    for g, sq in ((got_one, one), (got_two, two), (got_longer, longer), (got_three, three)):
        c = _cos(g, bert_ref.pool(bert_ref.forward_plain(d, sq), "mean"))
        assert np.all(c >= 0.999), c


def test_query_forward_switches(gpu, tmp_path_factory, monkeypatch):
    """The one-query forward's end-of-round-4 kernels against the ones they replace, same encoder, same inputs (the
    switches are read per launch): the exact-width reduce + LayerNorm kernel changes no bit (same arithmetic in the same
    order; DPP / permlane butterflies in both); the attention inside the attention-output GEMM and the 16-wave
    LayerNorm-inside GEMM reorder fp32 partial sums (16 K slices instead of 4): embeddings agree like any two batchings
    of the same text do."""
    from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
    d = str(tmp_path_factory.mktemp("large_model_switches"))
    cfg = EncoderConfig(pooling="mean")
    write_random_model_dir(d, cfg, seed=5)
    rng = np.random.default_rng(17)
    cases = [[list(rng.integers(0, 30522, size=n)) for n in lens] for lens in ([12], [16], [24], [9, 7], [30], [12, 12, 12, 12])]
    enc = HipSentenceEncoder.from_dir(d, device=0)
    try:
        base = [enc.encode_ids(c) for c in cases]
        again = [enc.encode_ids(c) for c in cases]
        monkeypatch.setenv("RASS_LN_EXACT", "0")
        general_ln = [enc.encode_ids(c) for c in cases]
        monkeypatch.delenv("RASS_LN_EXACT")
        monkeypatch.setenv("RASS_ATTN_FUSE", "0")
        pair = [enc.encode_ids(c) for c in cases]
        monkeypatch.setenv("RASS_ATTN_FUSE", "2")
        forced = [enc.encode_ids(c) for c in cases]
        monkeypatch.delenv("RASS_ATTN_FUSE")
        monkeypatch.setenv("RASS_GEMM_LNIN_WAVES", "4")
        lnin4 = [enc.encode_ids(c) for c in cases]
    finally:
        enc.close()
    for b, a, g, p_, f, l4 in zip(base, again, general_ln, pair, forced, lnin4):
        assert np.all(np.isfinite(b))
        assert np.array_equal(b, a)                       # deterministic
        assert np.array_equal(b, g)                       # exact-width LayerNorm kernel: the same bits
        for other in (p_, f, l4):
            assert np.all(_cos(b, other) >= 0.9995), _cos(b, other)


def test_base_shape_encoder_through_the_persistent_gemm(gpu, tmp_path_factory):
    """BERT-base-class layers (768 x 12 heads x 3072, the bge-base / nomic class) with > 1024 packed
    tokens: every linear goes through the persistent 256 x 256 ring GEMM (N % 256 == 0, M >= 1024),
    attention through full 512-token and ragged sequences; pooled vectors vs the fp32 oracle."""
    from oracle import bert_ref
    from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
    d = str(tmp_path_factory.mktemp("base_model"))
    cfg = EncoderConfig(vocab_size=2000, hidden=768, layers=2, heads=12, intermediate=3072, max_positions=512,
                        pooling="mean")
    write_random_model_dir(d, cfg, seed=4)
    rng = np.random.default_rng(5)
    lens = (512, 300, 200, 129, 64, 17, 1)
    seqs = [list(rng.integers(0, 2000, size=n)) for n in lens]
    enc = HipSentenceEncoder.from_dir(d, device=0)
    try:
        got = enc.encode_ids(seqs)
    finally:
        enc.close()
    ref = bert_ref.pool(bert_ref.forward_plain(d, seqs), "mean")
    c = _cos(got, ref)
    assert got.shape == (len(lens), 768) and np.all(np.isfinite(got))
    assert np.all(c >= 0.999), c


@pytest.mark.parametrize("hidden,heads", [(1536, 24), (2048, 32)])
def test_wide_hidden_sizes(gpu, tmp_path_factory, hidden, heads):
    """hidden 1536 / 2048 (the widest the C ABI accepts): LayerNorm rows of 3 and 4 x 512 columns, 24 / 32 heads, both in
    the few-row path (one short sequence) and in the batch path (> 1024 packed tokens, long sequences)."""
    from oracle import bert_ref
    from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
    d = str(tmp_path_factory.mktemp(f"wide_{hidden}"))
    cfg = EncoderConfig(vocab_size=1000, hidden=hidden, layers=1, heads=heads, intermediate=512, max_positions=512,
                        pooling="mean")
    write_random_model_dir(d, cfg, seed=hidden)
    rng = np.random.default_rng(hidden)
    enc = HipSentenceEncoder.from_dir(d, device=0)
    try:
        for lens in ((9,), (512, 480, 400, 33)):
            seqs = [list(rng.integers(0, 1000, size=n)) for n in lens]
            got = enc.encode_ids(seqs)
            ref = bert_ref.pool(bert_ref.forward_plain(d, seqs), "mean")
            assert got.shape == (len(lens), hidden) and np.all(np.isfinite(got))
            assert np.all(_cos(got, ref) >= 0.999), (lens, _cos(got, ref))
    finally:
        enc.close()


def test_embedding_shim_over_hip_encoder(gpu, tiny_model):
    """embed_texts_in_batches / embed_query (reference app/main.py:240-274) over the HIP encoder."""
    import asyncio
    from rassengine_amd import embedding
    from rassengine_amd.encoder import HipSentenceEncoder
    enc = HipSentenceEncoder.from_dir(tiny_model, device=0)
    embedding.set_embedder(enc)
    try:
        texts = ["patient history of diabetes", "", "blood pressure note", "chunk number 3 about topic"]
        e = asyncio.run(embedding.embed_texts_in_batches(texts, batch_size=3))
        assert e.shape == (4, 128) and e.dtype == np.float32
        assert np.all(e[1] == 0) and np.all(np.isfinite(e))
        q = asyncio.run(embedding.embed_query("blood pressure note"))
        assert q.shape == (1, 128) and np.allclose(q[0], e[2], atol=1e-6)
    finally:
        embedding.set_embedder(None)
        enc.close()


def test_encoder_kernels_as_torch_library_ops(gpu, tmp_path):
    """north_star: the encoder's kernels "under PyTorch-ROCm custom ops" — torch.ops.rass.gemm_bf16 / attention_bf16 /
    encode on torch's current stream equal the direct C-ABI calls (same kernels) and a plain fp32 torch reference."""
    torch = gpu
    from rassengine_amd import ops  # noqa: F401
    from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    M, N, K = 300, 512, 256
    X = torch.randn((M, K), generator=g, device="cuda").bfloat16()
    W = (torch.randn((N, K), generator=g, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn((N,), generator=g, device="cuda") * 0.1
    R = torch.randn((M, N), generator=g, device="cuda").bfloat16()
    for epi, res in ((0, None), (1, R), (2, None)):
        y = torch.ops.rass.gemm_bf16(X, W, b, res, epi)
        ref = X.float() @ W.float().T + b
        ref = ref + R.float() if epi == 1 else (torch.nn.functional.gelu(ref) if epi == 2 else ref)
        assert y.shape == (M, N) and bool(((y.float() - ref).abs() <= 1.5 * 2.0 ** -8 * ref.abs() + 2e-3).all())
    heads, lens = 4, [70, 33, 128]
    qkv = (torch.randn((sum(lens), 3 * heads * 64), generator=g, device="cuda") * 0.7).bfloat16()
    cu = torch.tensor([0, 70, 103, 231], dtype=torch.int32, device="cuda")
    ctx = torch.ops.rass.attention_bf16(qkv, cu, 128, heads)
    t = 0
    for n in lens:
        blk = qkv[t:t + n].float().view(n, 3, heads, 64)
        q, k, v = blk[:, 0].transpose(0, 1), blk[:, 1].transpose(0, 1), blk[:, 2].transpose(0, 1)
        ref = (torch.softmax(q @ k.transpose(1, 2) / 8.0, dim=-1) @ v).transpose(0, 1).reshape(n, heads * 64)
        assert bool(((ctx[t:t + n].float() - ref).abs() <= 2.0 ** -7 * (ref.abs() + 1.0)).all())
        t += n
    d = str(tmp_path / "tiny")
    write_random_model_dir(d, EncoderConfig(vocab_size=300, hidden=128, layers=2, heads=2, intermediate=512, max_positions=512,
                                            pooling="mean"), seed=5)
    enc = HipSentenceEncoder.from_dir(d, device=0)
    try:
        rng = np.random.default_rng(2)
        seqs = [list(rng.integers(0, 300, size=n)) for n in (9, 64, 130)]
        want = enc.encode_ids(seqs)
        ids = torch.tensor([t_ for s in seqs for t_ in s], dtype=torch.int32, device="cuda")
        cu2 = torch.tensor([0, 9, 73, 203], dtype=torch.int32, device="cuda")
        got = torch.ops.rass.encode(enc.handle, ids, cu2, 130, 128)
        torch.cuda.synchronize()
        assert got.shape == (3, 128) and np.array_equal(got.cpu().numpy(), want)      # the same forward, bit for bit
    finally:
        enc.close()
