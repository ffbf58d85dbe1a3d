import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu():
    """torch + a visible GPU; GPU tests fail (not skip) when the HIP library is missing."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import rassengine_amd._native as N
    N.lib()  # raises loudly if librass_hip.so is absent
    return torch


@pytest.fixture(scope="session")
def large_model(gpu, tmp_path_factory):
    """(model dir, HipSentenceEncoder) of the BERT-large-class shape (24 x 1024 x 16 heads x 4096, mean pooling) with the
    seed of the committed encoder fixtures: built once per session (334 M seeded weights take ~15 s to write and load)."""
    import numpy as np
    from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, write_random_model_dir
    fx = np.load(os.path.join(ROOT, "tests", "golden", "encoder_large_S32_B2.npz"))
    d = str(tmp_path_factory.mktemp("large_model"))
    write_random_model_dir(d, EncoderConfig(pooling="mean"), seed=int(fx["seed"]))
    enc = HipSentenceEncoder.from_dir(d, device=0)
    yield d, enc
    enc.close()
