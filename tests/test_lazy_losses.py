"""The lazy losses `Algo.update(planned_batch)` returns (recnn_amd/nn/algo.py): dict of the reference's shape
({'value', 'policy', 'step'}, recnn/nn/update/ddpg.py:102-103) whose values resolve on first use."""
from recnn_amd.nn.algo import LazyLosses, PlannedBatch

import pytest


class _FakeAlgo:
    def __init__(self):
        self.reads = 0

    def _loss_of(self, step):
        self.reads += 1
        return {"value": 2.5 + step, "policy": -1.25}


def test_lazy_losses_resolve_once_and_behave_like_floats():
    algo = _FakeAlgo()
    lz = LazyLosses(algo, 7, ("value", "policy"))
    assert set(lz) == {"value", "policy", "step"} and lz["step"] == 7 and algo.reads == 0
    v = lz["value"]
    assert algo.reads == 0
    assert float(v) == 9.5 and v.item() == 9.5 and algo.reads == 1
    assert float(v) == 9.5 and algo.reads == 1                    # cached
    assert v + 1 == 10.5 and 1 + v == 10.5 and v * 2 == 19.0 and v / 2 == 4.75 and 19 / v == 2.0 and -v == -9.5
    assert v > 9 and v <= 9.5 and v == 9.5 and v != 3
    assert f"{lz['policy']:.2f}" == "-1.25" and abs(lz["policy"]) == 1.25
    assert sum([lz["value"], lz["policy"]]) == 8.25             # what a plotter's running mean does


def test_planned_batch_without_a_position_cannot_be_materialised():
    b = PlannedBatch(object(), 3)
    assert "3" in repr(b) and "state" in b
    with pytest.raises(TypeError, match="position"):
        b["state"]
