"""A short run of tests/soak.py: randomized SlamUpdate sequences whose maps evolve over many steps (births, merges,
MaxQuantity cuts, resampling), device against the oracle at every step."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_randomized_sequences_follow_the_oracle():
    import soak
    worst = soak.run(4, 8, log=lambda *a: None)
    assert worst < 1e-6
