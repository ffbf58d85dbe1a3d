"""CPU tests around the encoder: the WordPiece tokenizer, the seeded model-dir writer and
the encoder oracle (hand-written fp32 forward == Hugging Face BertModel)."""
import numpy as np
import pytest

from rassengine_amd.encoder import EncoderConfig, WordPieceTokenizer, load_weights, weight_names, \
    write_random_model_dir

TINY = EncoderConfig(vocab_size=300, hidden=128, layers=2, heads=2, intermediate=512, max_positions=64)


@pytest.fixture(scope="module")
def tiny_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("tiny_model"))
    write_random_model_dir(d, TINY, seed=7)
    return d


def test_model_dir_roundtrip_is_deterministic(tiny_dir, tmp_path):
    cfg = EncoderConfig.from_dir(tiny_dir)
    assert (cfg.hidden, cfg.layers, cfg.heads, cfg.intermediate, cfg.max_positions) == (128, 2, 2, 512, 64)
    assert cfg.pooling == "cls"
    w = load_weights(tiny_dir)
    assert sorted(w) == sorted(weight_names(2))
    assert w["encoder.layer.1.intermediate.dense.weight"].shape == (512, 128)
    assert w["encoder.layer.0.output.dense.weight"].shape == (128, 512)
    other = str(tmp_path / "again")
    write_random_model_dir(other, TINY, seed=7)
    w2 = load_weights(other)
    assert all(np.array_equal(w[k], w2[k]) for k in w)


def test_wordpiece_tokenizer_matches_bert_rules():
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "un", "##aff", "##able", "the", "patient", "has", "type",
             "2", "diabetes", ",", ".", "##s", "cafe", "中", "文", "hyper", "##tension", "!"]
    tok = WordPieceTokenizer(vocab)
    v = tok.vocab
    assert tok.encode("Unaffable") == [v["[CLS]"], v["un"], v["##aff"], v["##able"], v["[SEP]"]]
    assert tok.encode("The patient has Type 2 diabetes.") == \
        [2, v["the"], v["patient"], v["has"], v["type"], v["2"], v["diabetes"], v["."], 3]
    assert tok.encode("patients, hypertension!") == [2, v["patient"], v["##s"], v[","], v["hyper"], v["##tension"], v["!"], 3]
    assert tok.encode("Café") == [2, v["cafe"], 3]                    # lower-case + accent strip
    assert tok.encode("中文") == [2, v["中"], v["文"], 3]               # CJK chars split
    assert tok.encode("xyzzy the") == [2, v["[UNK]"], v["the"], 3]      # no piece -> whole word [UNK]
    assert tok.encode("") == [2, 3]
    assert tok.encode("the\tpatient\n has type") == [2, v["the"], v["patient"], v["has"], v["type"], 3]
    long = tok.encode(" ".join(["the"] * 1000), max_len=16)
    assert len(long) == 16 and long[0] == 2 and long[-1] == 3           # truncated incl. [CLS]/[SEP]


def test_tokenizer_agrees_with_hf_bert_tokenizer(tmp_path):
    transformers = pytest.importorskip("transformers")
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("abcdefghijklmnopqrstuvwxyz0123456789.,!?-()") + \
        ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"] + \
        ["the", "patient", "blood", "pressure", "##ure", "press", "diabet", "##es", "##ic", "mg", "dose", "of", "is"]
    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(vocab) + "\n", encoding="utf-8")
    ours = WordPieceTokenizer.from_file(str(p))
    hf = transformers.BertTokenizer(str(p), do_lower_case=True)
    for text in ["The patient's blood-pressure is 120/80 (mg).", "Diabetic dose of 5mg!!", "ÀÉÎõü naïve café",
                 "tabs\tand\nnewlines  and   spaces", "", "x" * 150 + " the", "pressure pressures press-ure"]:
        assert ours.encode(text, 64) == hf.encode(text, truncation=True, max_length=64), text


def test_oracle_plain_forward_equals_hf(tiny_dir):
    pytest.importorskip("transformers")
    from oracle import bert_ref
    rng = np.random.default_rng(0)
    seqs = [list(rng.integers(0, 300, size=n)) for n in (5, 17, 64, 1)]
    a = bert_ref.forward_plain(tiny_dir, seqs)
    b = bert_ref.forward_hf(tiny_dir, seqs)
    for x, y in zip(a, b):
        assert x.shape == y.shape
        np.testing.assert_allclose(x, y, rtol=2e-4, atol=2e-5)
    e = bert_ref.pool(a, "mean", normalize=True)
    assert e.shape == (4, 128) and np.allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-6)
    assert np.array_equal(bert_ref.pool(a, "cls")[1], a[1][0])


# ---- committed encoder fixtures (tests/golden/make_encoder_fixtures.py): the oracle must still produce them
def _unpack(ids, cu):
    return [ids[cu[i]:cu[i + 1]].tolist() for i in range(len(cu) - 1)]


def _probe_sha(model_dir):
    import hashlib
    w = load_weights(model_dir)["encoder.layer.0.intermediate.dense.weight"]
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(w).tobytes()).digest(), dtype=np.uint8)


def test_oracle_reproduces_tiny_encoder_fixture(tmp_path):
    import os
    from oracle import bert_ref
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_tiny_L2_H128.npz"))
    v, h, l, a, i, p = (int(x) for x in fx["config"])
    d = str(tmp_path / "tiny")
    write_random_model_dir(d, EncoderConfig(vocab_size=v, hidden=h, layers=l, heads=a, intermediate=i, max_positions=p),
                           seed=int(fx["seed"]))
    assert np.array_equal(_probe_sha(d), fx["weight_sha256"])      # the seeded weight generator has not drifted
    hid = bert_ref.forward_plain(d, _unpack(fx["token_ids"], fx["cu_seqlens"]))
    # fp32 CPU matmuls may block differently between hosts: allow a few ulp of drift, not more
    assert np.allclose(bert_ref.pool(hid, "cls"), fx["pooled_cls"], rtol=0, atol=2e-5)
    assert np.allclose(bert_ref.pool(hid, "mean"), fx["pooled_mean"], rtol=0, atol=2e-5)
    hf = bert_ref.forward_hf(d, _unpack(fx["token_ids"], fx["cu_seqlens"]))
    assert np.allclose(bert_ref.pool(hf, "mean"), fx["pooled_mean"], rtol=0, atol=5e-5)


def test_oracle_reproduces_large_encoder_fixture(tmp_path):
    """BERT-large-class shape: regenerates the 334 M seeded weights (~15 s) and checks the S=32, B=2 fixture
    and the first two chunks of the end-to-end corpus fixture."""
    import os
    from oracle import bert_ref
    g = os.path.join(os.path.dirname(__file__), "golden")
    fx = np.load(os.path.join(g, "encoder_large_S32_B2.npz"))
    d = str(tmp_path / "large")
    write_random_model_dir(d, EncoderConfig(pooling="mean"), seed=int(fx["seed"]))
    assert np.array_equal(_probe_sha(d), fx["weight_sha256"])
    hid = bert_ref.forward_plain(d, _unpack(fx["token_ids"], fx["cu_seqlens"]))
    assert np.allclose(bert_ref.pool(hid, "mean"), fx["pooled_mean"], rtol=0, atol=1e-4)
    assert np.allclose(bert_ref.pool(hid, "cls"), fx["pooled_cls"], rtol=0, atol=1e-4)
    e2e = np.load(os.path.join(g, "e2e_large_corpus.npz"))
    assert np.array_equal(e2e["weight_sha256"], fx["weight_sha256"])
    docs = _unpack(e2e["doc_token_ids"], e2e["doc_cu_seqlens"])
    got = bert_ref.pool(bert_ref.forward_plain(d, docs[:2]), "mean")
    assert np.allclose(got, e2e["doc_embeddings"][:2], rtol=0, atol=1e-4)
    # the stored top-5 is the fp64 ranking of the stored embeddings
    from oracle import oracle as O
    xn = O.normalize_ref(e2e["doc_embeddings"]).astype(np.float32)
    qn = O.normalize_ref(e2e["query_embeddings"]).astype(np.float32)
    s, i = O.search(xn, qn, 5, kind=O.KIND_F64)
    assert np.array_equal(i, e2e["top5_ids"]) and np.abs(s - e2e["top5_scores"]).max() < 1e-12
    assert all(len(set(e2e["doc_family"][row])) == 1 for row in i)  # every query's top-5 is one topic family
