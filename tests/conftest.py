import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _limit_torch_threads():
    # The oracle runs small fp32 convolutions; on a many-core host (the MI355X box has 256 logical CPUs) torch's
    # default of one thread per core makes them several times SLOWER.  16 threads keeps both suites short.
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except Exception:
        pass


_limit_torch_threads()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "grad: the test needs autograd enabled (training path)")


@pytest.fixture(autouse=True)
def _inference_by_default(request):
    """The reference's inference entry points run under torch.inference_mode (models/model.py:59, flow_matching.py:24);
    parity tests therefore run with autograd off unless they are marked ``grad`` (with autograd on, a module whose
    parameters require grad takes the training path: native forward that keeps activations + native backward)."""
    import torch
    if request.node.get_closest_marker("grad") or not request.node.get_closest_marker("gpu"):
        yield
    else:
        with torch.no_grad():
            yield


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
