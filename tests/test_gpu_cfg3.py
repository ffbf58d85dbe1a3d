"""BASELINE cfg 3 on the GPU ("mxbai-embed-large bf16 batch ingest + top-10 search, end-to-end embed+index"):
the BERT-large-class encoder (24 x 1024 x 16 heads x 4096, seeded random weights — no real weights exist
offline) through the C ABI, at full 512-token windows and ragged lengths, against

  * the committed encoder fixtures (tests/golden/make_encoder_fixtures.py, outputs of oracle/bert_ref.py),
  * the fp32 CPU oracle run here on a fresh batch,
  * the device-resident hand-off rass_encode_device -> rass_index_add_device (K8, the ingest path of
    scripts/bench_ingest.py) against the host path rass_encode -> rass_index_add, bit for bit,
  * SURVEY §8c O3 end to end: HIP-encoder + HIP-search vs oracle-encoder + oracle-search on a fixture corpus.

Tolerances (written here, north_star / SURVEY §8c): pooled sentence vector cosine >= 0.999 to the fp32
oracle (bf16 weights AND bf16 activations on the GPU); search over the SAME embeddings within 2e-6 of fp64
(north_star: 1e-3); end to end |d cos| on returned scores is bounded by the encoder's bf16 error, see
test_end_to_end_embed_index_search.
"""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _cos(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return np.sum(a * b, axis=-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def _unpack(ids, cu):
    return [ids[cu[i]:cu[i + 1]].tolist() for i in range(len(cu) - 1)]


@pytest.fixture(scope="module")
def large(large_model):
    """(model dir, HipSentenceEncoder) of the BERT-large-class shape, the seed of the committed fixtures."""
    return large_model


def test_large_fixture_S32_B2(large):
    _, enc = large
    fx = np.load(os.path.join(GOLDEN, "encoder_large_S32_B2.npz"))
    got = enc.encode_ids(_unpack(fx["token_ids"], fx["cu_seqlens"]))
    assert got.shape == (2, 1024) and got.dtype == np.float32
    assert np.all(_cos(got, fx["pooled_mean"]) >= 0.999)


@pytest.mark.parametrize("pooling", ["cls", "mean"])
def test_tiny_fixture(gpu, tmp_path, pooling, monkeypatch):
    from rassengine_amd import config
    from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
    fx = np.load(os.path.join(GOLDEN, "encoder_tiny_L2_H128.npz"))
    v, h, l, a, i, p = (int(x) for x in fx["config"])
    d = str(tmp_path / "tiny")
    write_random_model_dir(d, EncoderConfig(vocab_size=v, hidden=h, layers=l, heads=a, intermediate=i, max_positions=p),
                           seed=int(fx["seed"]))
    monkeypatch.setattr(config, "RASS_POOLING", pooling)
    enc = HipSentenceEncoder.from_dir(d, device=0)
    try:
        got = enc.encode_ids(_unpack(fx["token_ids"], fx["cu_seqlens"]))
    finally:
        enc.close()
    assert np.all(_cos(got, fx["pooled_" + pooling]) >= 0.999)


def test_full_window_batch_matches_oracle(large):
    """>= 4 sequences of 512 tokens plus ragged ones (the verdict's cfg-3 shape): 2 600 packed tokens go
    through the persistent 256x256 GEMM, the 512-key attention and the varlen tails."""
    from oracle import bert_ref
    d, enc = large
    rng = np.random.default_rng(512)
    lens = [512, 512, 512, 512, 300, 129, 64, 17, 2, 1]
    seqs = [list(rng.integers(0, 30522, size=n)) for n in lens]
    got = enc.encode_ids(seqs)
    got_rev = enc.encode_ids(seqs[::-1])[::-1]
    ref = bert_ref.pool(bert_ref.forward_plain(d, seqs), "mean")
    c = _cos(got, ref)
    assert np.all(np.isfinite(got)) and np.all(c >= 0.999), c
    assert np.array_equal(got, got_rev)  # a sequence's embedding does not depend on its batch neighbours


def _encode_device(enc, seqs, torch):
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    cu = np.zeros(len(seqs) + 1, dtype=np.int32)
    np.cumsum(lens, out=cu[1:])
    ids = np.concatenate([np.asarray(s, dtype=np.int32) for s in seqs])
    d_ids = torch.from_numpy(ids).cuda()
    d_cu = torch.from_numpy(cu).cuda()
    out = torch.zeros((len(seqs), enc.dim), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()                       # inputs were staged on torch's stream, the encoder has its own
    enc.encode_device(d_ids.data_ptr(), d_cu.data_ptr(), len(seqs), int(cu[-1]), int(lens.max()), out.data_ptr())
    return out, (d_ids, d_cu)


def test_device_handoff_equals_host_path(large):
    """K8: rass_encode_device -> rass_index_add_device (no host round trip) must store the very rows, and
    answer searches with the very bits, of rass_encode -> rass_index_add."""
    import torch
    from rassengine_amd.engine import Engine
    _, enc = large
    rng = np.random.default_rng(8)
    lens = [512, 400, 257, 256, 255, 128, 100, 64, 33, 16, 9, 3] * 4      # 48 chunks, 8 132 tokens
    seqs = [list(rng.integers(0, 30522, size=n)) for n in lens]
    tags = (np.arange(len(seqs)) % 3 + 1).astype(np.int32)
    eng = Engine(0, 1024)
    try:
        host_idx = eng.open_index("cfg3-host")
        dev_idx = eng.open_index("cfg3-dev")
        emb = enc.encode_ids(seqs)                                    # rass_encode: host ids in, host fp32 out
        assert host_idx.add(emb, tags=tags, normalize=True) == 0       # rass_index_add
        # encoder + index on ONE stream (the encoder's own): the add is ordered behind the forward with no
        # host synchronisation in between.  (torch's default stream is HIP's null stream, pointer 0, which
        # rass_encode_device reads as "my own stream" — the two would NOT be ordered; see the header.)
        assert enc.stream != 0
        eng.set_stream(enc.stream)
        out, keep = _encode_device(enc, seqs, torch)
        d_tags = torch.from_numpy(tags).cuda()
        assert dev_idx.add_device(out.data_ptr(), len(seqs), d_tags_ptr=d_tags.data_ptr(), normalize=True) == 0
        torch.cuda.synchronize()
        eng.reset_stream()
        assert np.array_equal(out.cpu().numpy(), emb)                  # same forward, same bits
        assert dev_idx.count == host_idx.count == len(seqs)
        assert np.array_equal(dev_idx.get_rows(0, len(seqs)), host_idx.get_rows(0, len(seqs)))
        q = emb[::5] + 0.05 * rng.standard_normal(emb[::5].shape).astype(np.float32)
        qf = np.array([-1, 1, 2, 3] * 3, dtype=np.int32)[:q.shape[0]]
        s_h, i_h = host_idx.search(q, 10, q_filter=qf)
        s_d, i_d = dev_idx.search(q, 10, q_filter=qf)
        assert np.array_equal(i_h, i_d) and np.array_equal(s_h, s_d)
        # a second, differently composed device batch appended behind the first: rows land at the right ids
        eng.set_stream(enc.stream)
        out2, keep2 = _encode_device(enc, seqs[:5], torch)
        first = dev_idx.add_device(out2.data_ptr(), 5, normalize=True)
        torch.cuda.synchronize()
        eng.reset_stream()
        assert first == len(seqs) and dev_idx.rows == len(seqs) + 5
        # the right rows at the right ids.  Not bit for bit: which GEMM / attention kernel runs depends on the batch (rows,
        # mean sequence length), so the same text embedded in another batch agrees to the kernels' tolerance, not to the
        # bit; the SAME batch gives the same bits (asserted above and in test_gpu_encoder)
        a, b = dev_idx.get_rows(first, 5), host_idx.get_rows(0, 5)
        cos = np.sum(a.astype(np.float64) * b, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
        assert np.all(cos >= 0.9999), cos
    finally:
        eng.close()


def _ingest_batch_checks(large, torch, seqs, sample):
    """One ingest batch at cfg 3's shape through BOTH paths: rass_encode -> rass_index_add (host) and rass_encode_device ->
    rass_index_add_device (K8, no host round trip).  Every stored row finite and unit-norm, the two paths' rows and
    embeddings bit-equal, `sample` sequences >= 0.999 cosine to the fp32 CPU oracle, and the same `sample` sequences
    embedded as a batch of their own within the kernels' tolerance (cosine >= 0.9998: other GEMM / attention kernels run
    for 4 sequences than for 256 — and the LayerNorm-folded GEMM epilogues from 12 288 tokens on)."""
    from oracle import bert_ref
    from rassengine_amd.engine import Engine
    d, enc = large
    n = len(seqs)
    s0 = enc.stats()
    emb = enc.encode_ids(seqs)
    s1 = enc.stats()
    assert s1["forwards"] - s0["forwards"] == 1 and s1["tokens"] - s0["tokens"] == sum(len(s) for s in seqs)
    assert emb.shape == (n, 1024) and np.all(np.isfinite(emb))
    eng = Engine(0, 1024)
    try:
        host_idx, dev_idx = eng.open_index("cfg3-full-host"), eng.open_index("cfg3-full-dev")
        assert host_idx.add(emb, normalize=True) == 0
        eng.set_stream(enc.stream)
        out, keep = _encode_device(enc, seqs, torch)
        assert dev_idx.add_device(out.data_ptr(), n, normalize=True) == 0
        torch.cuda.synchronize()
        eng.reset_stream()
        assert np.array_equal(out.cpu().numpy(), emb)                     # the same batch gives the same bits
        rows = dev_idx.get_rows(0, n)
        assert np.array_equal(rows, host_idx.get_rows(0, n))              # device hand-off == host path
        assert np.all(np.isfinite(rows)) and np.abs(np.linalg.norm(rows.astype(np.float64), axis=1) - 1.0).max() < 1e-5
        s, i = dev_idx.search(emb[sample], 1)
        assert np.array_equal(i[:, 0], np.asarray(sample)) and np.all(s[:, 0] > 0.99999)
    finally:
        eng.close()
    sub = [seqs[j] for j in sample]
    ref = bert_ref.pool(bert_ref.forward_plain(d, sub), "mean")
    c_ref = _cos(emb[sample], ref)
    alone = enc.encode_ids(sub)
    c_alone = _cos(emb[sample], alone)
    print(f"cfg-3 batch of {n} sequences / {sum(len(s) for s in seqs)} tokens: cosine vs fp32 oracle >= {c_ref.min():.6f}, "
          f"vs the same sequences in a batch of {len(sub)} >= {c_alone.min():.7f}")
    assert np.all(c_ref >= 0.999), c_ref
    # 0.9998: since round 4 a batch of >= 12 288 tokens runs with both LayerNorms of every layer folded into the GEMMs around
    # them (weights pre-scaled by gamma and re-rounded to bf16, csrc/encoder_gemm.hip LnFold), a batch of 4 sequences the
    # unfused kernels: two bf16 evaluations of the same fp32 function, each 0.99987 from the fp32 oracle over 24 layers and
    # 0.99989 from each other (0.99994 between the big and the small unfused paths); tests/test_gpu_ln_fold.py pins the pair
    assert np.all(c_alone >= 0.9998), c_alone


def test_cfg3_full_batch_256x512(large):
    """BASELINE cfg 3 at its stated shape: ONE forward of 256 sequences x 512 tokens = 131 072 packed tokens (the 256^2
    persistent GEMM on 512 row tiles, attention64_kernel on 4 096 (sequence, head) items)."""
    import torch
    rng = np.random.default_rng(99)
    seqs = [rng.integers(0, 30522, size=512).tolist() for _ in range(256)]
    _ingest_batch_checks(large, torch, seqs, sample=[0, 85, 170, 255])


def test_cfg3_ragged_batch_256(large):
    """256 sequences with lengths uniform in [64, 512] (varlen packing, the second length profile of SURVEY §8d cfg 3)."""
    import torch
    rng = np.random.default_rng(100)
    lens = rng.integers(64, 513, size=256)
    lens[[3, 77]] = [64, 512]
    seqs = [rng.integers(0, 30522, size=int(n)).tolist() for n in lens]
    _ingest_batch_checks(large, torch, seqs, sample=[3, 77, 128, 254])


def test_end_to_end_embed_index_search(large, oracle):
    """SURVEY §8c O3 on the committed fixture corpus (60 chunks in 10 topic families of 24..512 tokens, 8
    queries): HIP encoder -> HIP index -> HIP search against oracle encoder (fp32 CPU) -> oracle search (fp64).

    What is asserted, and why these numbers:
      * search exactness: over the HIP encoder's OWN embeddings the HIP search returns the fp64 oracle's ids
        and scores within 2e-6 (north_star asks 1e-3);
      * encoder: cosine >= 0.999 per chunk / query vs the oracle embedding;
      * end to end: every query's top-5 is the oracle's topic family (6 members; ranks INSIDE a family are
        separated by 1e-4..1e-3 in cosine, below the bf16 encoder's resolution) so id overlap >= 4/5, and the
        returned score of a chunk differs from the oracle's score for the SAME chunk by <= E2E_DCOS.
    E2E_DCOS = 2e-3 is the bound of the bf16 (weights + activations) encoder, not of the search: the measured
    maximum is printed; the fp32 search on top of it contributes < 2e-6.
    """
    from rassengine_amd.engine import Engine
    E2E_DCOS = 2e-3
    _, enc = large
    fx = np.load(os.path.join(GOLDEN, "e2e_large_corpus.npz"))
    docs = _unpack(fx["doc_token_ids"], fx["doc_cu_seqlens"])
    queries = _unpack(fx["query_token_ids"], fx["query_cu_seqlens"])
    e_docs = enc.encode_ids(docs)
    e_q = enc.encode_ids(queries)
    assert np.all(_cos(e_docs, fx["doc_embeddings"]) >= 0.999)
    assert np.all(_cos(e_q, fx["query_embeddings"]) >= 0.999)
    eng = Engine(0, 1024)
    try:
        idx = eng.open_index("cfg3-e2e")
        idx.add(e_docs, normalize=True)
        s, i = idx.search(e_q, 5)
    finally:
        eng.close()
    # (1) the search is exact over the embeddings it was given
    xn = oracle.normalize_ref(e_docs).astype(np.float32)
    qn = oracle.normalize_ref(e_q).astype(np.float32)
    rs, ri = oracle.search(xn, qn, 5, kind=oracle.KIND_F64)
    assert np.array_equal(i, ri)
    assert np.abs(s.astype(np.float64) - rs).max() <= 2e-6
    # (2) end to end against oracle-encoder + oracle-search
    fam = fx["doc_family"]
    o_ids, o_scores = fx["top5_ids"], fx["top5_scores"]
    xo = oracle.normalize_ref(fx["doc_embeddings"]).astype(np.float64)
    qo = oracle.normalize_ref(fx["query_embeddings"]).astype(np.float64)
    worst = 0.0
    for r in range(len(queries)):
        assert set(fam[i[r]]) == set(fam[o_ids[r]]) and len(set(fam[i[r]])) == 1
        assert len(set(i[r].tolist()) & set(o_ids[r].tolist())) >= 4
        same_chunk = np.abs(s[r].astype(np.float64) - xo[i[r]] @ qo[r])
        worst = max(worst, float(same_chunk.max()))
    print(f"end-to-end max |d cos| on returned chunks = {worst:.2e}; top-1 score gap vs oracle = "
          f"{np.abs(s[:, 0] - o_scores[:, 0]).max():.2e}")
    assert worst <= E2E_DCOS


def test_failed_workspace_grow_then_small_encode(large):
    """a failed workspace allocation must not leave a stale capacity behind (the next small
    forward used to run on freed / NULL activations)."""
    import torch
    from rassengine_amd import _native as N
    _, enc = large
    L = N.lib()
    ids = torch.zeros((8,), dtype=torch.int32, device="cuda")
    cu = torch.tensor([0, 8], dtype=torch.int32, device="cuda")
    out = torch.empty((1, 1024), dtype=torch.float32, device="cuda")
    rc = L.rass_encode_device(enc._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(cu.data_ptr()), 1,
                              2 ** 31 - 512, 8, ctypes.c_void_p(out.data_ptr()), None)   # ~4 TB of activations
    assert rc == -3, (rc, L.rass_last_error())  # RASS_ERR_OOM
    rng = np.random.default_rng(1)
    seqs = [list(rng.integers(0, 30522, size=n)) for n in (5, 40)]
    a = enc.encode_ids(seqs)
    b = enc.encode_ids(seqs)
    assert np.all(np.isfinite(a)) and np.array_equal(a, b)
