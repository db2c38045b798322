import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no GPU is visible")
    if os.environ.get("RECNN_LD_PAD"):          # run the GPU suite on a different leading-dimension padding (tuning knob)
        from recnn_amd import _lib as L
        L.load().recnn_tune_ld_pad(int(os.environ["RECNN_LD_PAD"]))
    return torch.device("cuda")


@pytest.fixture(autouse=True)
def _adam_is_the_optimizer_of_record():
    """The facades default to Ranger like the reference (recnn/nn/algo.py:84-89); the parity oracle and north_star's
    optimizer is Adam, so tests construct DDPG / TD3 with Adam unless they ask for Ranger themselves."""
    from recnn_amd.nn import algo
    algo.set_default_optimizer("adam")
    yield
    algo.set_default_optimizer("ranger")
