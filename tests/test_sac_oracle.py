"""oracle/sac_oracle.py against the golden runs of the reference notebook's own SAC cells (tests/golden/sac_*.npz, made by
oracle/make_golden_sac.py from `examples/1. Vanilla RL/4. SAC.ipynb` cells 5-8): losses and all four networks at 5e-5."""
import os

import numpy as np
import pytest

from tests import sac_replay as SR
from tests.helpers import rel_err

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["sac_small", "sac_wd"])
def test_sac_oracle_replays_the_notebook_run(name):
    fx = SR.load(os.path.join(GOLDEN, name + ".npz"))
    losses, final = SR.replay_oracle(fx)
    ref = fx["g"]["losses"]
    assert losses.shape == ref.shape == (fx["steps"], 4)
    assert rel_err(losses[:, 1:], ref[:, 1:]) < 5e-5
    for tag, p in final.items():
        for k, v in p.items():
            assert rel_err(v, fx["g"][f"final.{tag}.{k}"]) < 5e-5, (tag, k)
    # the quirks are visible in the fixture: one z per step, the value target is [B, A]-shaped (value loss != softq-style MSE)
    assert fx["g"]["z"].shape == (fx["steps"],)
