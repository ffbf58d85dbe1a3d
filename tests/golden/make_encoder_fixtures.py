"""Generates the encoder golden fixtures of SURVEY §8c (O3):

    tests/golden/encoder_tiny_L2_H128.npz    tiny model (2 x 128 x 2 heads x 512), 10 ragged sequences
    tests/golden/encoder_large_S32_B2.npz    BERT-large-class (24 x 1024 x 16 x 4096), 2 sequences of 32 tokens
    tests/golden/e2e_large_corpus.npz        BERT-large-class, 60 chunks + 8 queries (topic families with token
                                             substitutions, 24..512 tokens): oracle embeddings and the oracle's
                                             exact top-5 (BASELINE cfg 3 end to end: embed + index + search)

Run from the repo root:  python tests/golden/make_encoder_fixtures.py      (~2 min on 8 cores)

Expected outputs come from the CPU encoder oracle (oracle/bert_ref.py: plain fp32 PyTorch forward,
cross-checked against Hugging Face BertModel in tests/test_encoder_cpu.py) and the fp64 search oracle
(oracle/rass_oracle.c).  Weights are NOT stored: `write_random_model_dir(cfg, seed)` regenerates them
bit-identically on every host (numpy PCG64); each fixture stores the sha256 of one weight matrix so a
drift of the generator is caught.  The reference holds no embedding to pin these against
(tests/test_main.py:26; no weights offline): they pin OUR oracle — "parity unpinned", DESIGN.md §5.
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bert_ref  # noqa: E402
from oracle import oracle as O  # noqa: E402
from rassengine_amd.encoder import EncoderConfig, load_weights, write_random_model_dir  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

TINY = dict(vocab_size=300, hidden=128, layers=2, heads=2, intermediate=512, max_positions=512)
TINY_SEED = 11
LARGE_SEED = 1
CLS, SEP = 101, 102
PROBE_WEIGHT = "encoder.layer.0.intermediate.dense.weight"


def weight_sha(model_dir: str) -> np.ndarray:
    w = load_weights(model_dir)[PROBE_WEIGHT]
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(w).tobytes()).digest(), dtype=np.uint8)


def pack(seqs):
    cu = np.zeros(len(seqs) + 1, dtype=np.int32)
    np.cumsum([len(s) for s in seqs], out=cu[1:])
    return np.concatenate([np.asarray(s, dtype=np.int32) for s in seqs]), cu


def e2e_sequences(rng):
    """10 topic families x 6 variants (10 % of the tokens substituted) + 8 queries (20 % substituted)."""
    lens = [24, 40, 64, 96, 128, 160, 200, 300, 400, 512]
    docs, queries, family = [], [], []
    bases = [rng.integers(1000, 30000, size=n - 2) for n in lens]
    for t, base in enumerate(bases):
        for _ in range(6):
            v = base.copy()
            sub = rng.random(v.shape[0]) < 0.10
            v[sub] = rng.integers(1000, 30000, size=int(sub.sum()))
            docs.append([CLS] + v.tolist() + [SEP])
            family.append(t)
    for t in (0, 2, 3, 5, 6, 7, 8, 9):
        v = bases[t].copy()
        sub = rng.random(v.shape[0]) < 0.20
        v[sub] = rng.integers(1000, 30000, size=int(sub.sum()))
        queries.append([CLS] + v.tolist() + [SEP])
    return docs, queries, np.array(family, dtype=np.int32)


def main():
    with tempfile.TemporaryDirectory() as d:
        write_random_model_dir(d, EncoderConfig(**TINY), seed=TINY_SEED)
        rng = np.random.default_rng(3)
        seqs = [list(rng.integers(0, 300, size=n)) for n in (1, 2, 16, 17, 63, 64, 65, 129, 300, 512)]
        hid = bert_ref.forward_plain(d, seqs)
        ids, cu = pack(seqs)
        np.savez_compressed(os.path.join(HERE, "encoder_tiny_L2_H128.npz"), token_ids=ids, cu_seqlens=cu,
                            pooled_cls=bert_ref.pool(hid, "cls"), pooled_mean=bert_ref.pool(hid, "mean"),
                            seed=np.int64(TINY_SEED), weight_sha256=weight_sha(d),
                            config=np.array([TINY[k] for k in ("vocab_size", "hidden", "layers", "heads",
                                                               "intermediate", "max_positions")], dtype=np.int64))
    with tempfile.TemporaryDirectory() as d:
        write_random_model_dir(d, EncoderConfig(pooling="mean"), seed=LARGE_SEED)
        sha = weight_sha(d)
        rng = np.random.default_rng(99)
        seqs = [list(rng.integers(0, 30522, size=32)) for _ in range(2)]
        hid = bert_ref.forward_plain(d, seqs)
        ids, cu = pack(seqs)
        np.savez_compressed(os.path.join(HERE, "encoder_large_S32_B2.npz"), token_ids=ids, cu_seqlens=cu,
                            pooled_cls=bert_ref.pool(hid, "cls"), pooled_mean=bert_ref.pool(hid, "mean"),
                            seed=np.int64(LARGE_SEED), weight_sha256=sha)

        rng = np.random.default_rng(2024)
        docs, queries, family = e2e_sequences(rng)
        e_docs = bert_ref.pool(bert_ref.forward_plain(d, docs), "mean")
        e_q = bert_ref.pool(bert_ref.forward_plain(d, queries), "mean")
        xn = O.normalize_ref(e_docs).astype(np.float32)
        qn = O.normalize_ref(e_q).astype(np.float32)
        s, i = O.search(xn, qn, 5, kind=O.KIND_F64)
        d_ids, d_cu = pack(docs)
        q_ids, q_cu = pack(queries)
        np.savez_compressed(os.path.join(HERE, "e2e_large_corpus.npz"), doc_token_ids=d_ids, doc_cu_seqlens=d_cu,
                            query_token_ids=q_ids, query_cu_seqlens=q_cu, doc_family=family,
                            doc_embeddings=e_docs, query_embeddings=e_q, top5_ids=i, top5_scores=s,
                            seed=np.int64(LARGE_SEED), weight_sha256=sha)
        print("e2e oracle top-5 families:", family[i].tolist())
        print("e2e oracle top-5 scores:", np.round(s, 4).tolist())
    print("wrote encoder fixtures")


if __name__ == "__main__":
    main()
