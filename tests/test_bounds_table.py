"""The bf16 parity bounds (tests/helpers.py BF16_BOUNDS) against the values measured on an MI355X
(profiles/r03_measured_bounds.json, profiles/r05_measured_bounds.json for the quantities added in round 5; written by tests.helpers.within during `pytest -m gpu`): no bound looser than 3x measured."""
import json
import os

from tests.helpers import BF16_BOUNDS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16_bounds_are_within_3x_of_measured():
    rec = json.load(open(os.path.join(ROOT, "profiles", "r03_measured_bounds.json")))
    rec.update({k: v for k, v in json.load(open(os.path.join(ROOT, "profiles", "r05_measured_bounds.json"))).items() if k not in rec})   # (quantities new in round 5)
    assert BF16_BOUNDS, "empty bounds table"
    for name, bound in BF16_BOUNDS.items():
        assert name in rec, f"{name}: no measurement on record"
        measured = rec[name]["measured"]
        assert measured <= bound, (name, measured, bound)
        assert bound <= max(3.0 * measured, 1e-6), (name, measured, bound)
