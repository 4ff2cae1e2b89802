import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "reference_outputs.npz")
    return dict(np.load(path))


@pytest.fixture(scope="session")
def sd():
    from oracle import make_state_dict
    return make_state_dict(1234)


@pytest.fixture(scope="session")
def cfg_params():
    from oracle import make_cfg_params
    return make_cfg_params(4321)
