"""GPU: randomised parity sweep of the fused schedules (tools/fuzz_lanes.py) with a fixed seed: every plane size of the
register-resident families, channel counts that select every workgroup width, odd batch sizes, launch-knob overrides."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_configurations_match_the_c_oracle():
    import fuzz_lanes
    assert fuzz_lanes.run(60, seed=2024, verbose=False) == 0
