"""The big-batch forward with both LayerNorms of every layer folded into the GEMMs around them (csrc/encoder_gemm.hip, LnFold;
VERDICT r3 #3a): QKV / FFN-up run on the RAW residual sums with weights pre-scaled by gamma and apply (mean, rstd) + a rank-1
correction in their epilogues, attn-out / FFN-down rebuild the normalised residual element by element and emit the rows'
partial (sum, sum of squares).  Batches of >= 12 288 tokens of a 1024-wide model take that path (all four GEMMs on the
persistent 256^2 kernel); RASS_ENCODER_LN_FOLD=0 is the unfused pair (GEMM, then the stand-alone LayerNorm kernel).

Pinned: both paths against the fp32 CPU oracle of the same model (oracle/bert_ref.py, plain PyTorch fp32) at cosine >= 0.999
per sequence, and against each other far tighter than either is to the oracle; deterministic from run to run (no atomics)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cos(a, b):
    return np.sum(a * b, axis=-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


@pytest.mark.parametrize("pooling", ["mean", "cls"])
def test_folded_layernorm_forward_matches_oracle_and_unfused_path(gpu, tmp_path, pooling):
    from oracle import bert_ref
    from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
    cfg = EncoderConfig(vocab_size=2000, hidden=1024, layers=2, heads=16, intermediate=4096, max_positions=512, pooling=pooling)
    d = str(tmp_path / "m")
    write_random_model_dir(d, cfg, seed=21)
    rng = np.random.default_rng(4)
    lens = [512, 500, 449, 512, 300, 512, 77, 512, 512, 460, 512, 512, 391, 512, 512, 512, 205, 512, 512, 512, 512, 480,
            512, 512, 512, 512, 512, 401]
    assert sum(lens) >= 12288 + 256                       # the fold path's threshold, ragged last tile
    seqs = [list(rng.integers(0, 2000, size=n)) for n in lens]
    enc = HipSentenceEncoder.from_dir(d, device=0)
    old = os.environ.pop("RASS_ENCODER_LN_FOLD", None)
    try:
        fold = enc.encode_ids(seqs)
        fold2 = enc.encode_ids(seqs)
        os.environ["RASS_ENCODER_LN_FOLD"] = "0"
        plain = enc.encode_ids(seqs)
        os.environ.pop("RASS_ENCODER_LN_FOLD")
        small = enc.encode_ids(seqs[:3])                  # 1 461 tokens: below the threshold, the unfused kernels
    finally:
        os.environ.pop("RASS_ENCODER_LN_FOLD", None)
        if old is not None:
            os.environ["RASS_ENCODER_LN_FOLD"] = old
        enc.close()
    assert np.array_equal(fold, fold2)                    # deterministic: fixed summation order, no atomics
    assert np.all(np.isfinite(fold)) and fold.shape == (len(lens), 1024)
    ref = bert_ref.pool(bert_ref.forward_plain(d, seqs), pooling)
    c_fold, c_plain, c_both = _cos(fold, ref), _cos(plain, ref), _cos(fold, plain)
    print(f"pooling {pooling}: cosine to the fp32 oracle folded {c_fold.min():.6f} / unfused {c_plain.min():.6f}; "
          f"folded vs unfused {c_both.min():.7f}")
    assert c_fold.min() >= 0.999 and c_plain.min() >= 0.999
    assert c_both.min() >= 0.9999
    # the folded path is not LESS accurate than the unfused one (it normalises in fp32 after the matmul)
    assert (1 - c_fold).max() <= 2.0 * (1 - c_plain).max() + 1e-6
    assert _cos(small, plain[:3]).min() >= 0.9999
