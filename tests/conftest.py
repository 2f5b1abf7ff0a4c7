import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "custom_hmm_golden.npz")
    return np.load(path)


@pytest.fixture(scope="session")
def feature_set():
    from tests._synth import VOCAB, synth_feature_set
    by_word, flat = synth_feature_set(VOCAB, 6, D=13, seed=0)
    return by_word, flat
