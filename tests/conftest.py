import os
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_golden.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(GOLDEN))


@pytest.fixture(scope="session")
def synth_model():
    from smalify_amd import synthetic
    return synthetic.synthetic_model(seed=0, shape_family_id=1)


@pytest.fixture(scope="session")
def synth_model_family0():
    from smalify_amd import synthetic
    return synthetic.synthetic_model(seed=0, shape_family_id=0)
