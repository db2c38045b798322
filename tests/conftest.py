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
    # run the GPU suite under other schedules / tile shapes (tuning knobs never change results): every RECNN_* knob that is
    # set in the environment (recnn_amd/_tune.py), e.g. RECNN_SPLIT_FWD=0 python -m pytest tests -m gpu
    from recnn_amd._tune import apply_env_knobs
    apply_env_knobs()
    return torch.device("cuda")


@pytest.fixture(autouse=True)
def _adam_is_the_optimizer_of_record():
    """The facades default to Ranger like the reference (recnn/nn/algo.py:84-89); the parity oracle and north_star's
    optimizer is Adam, so tests construct DDPG / TD3 with Adam unless they ask for Ranger themselves."""
    from recnn_amd.nn import algo
    algo.set_default_optimizer("adam")
    yield
    algo.set_default_optimizer("ranger")
