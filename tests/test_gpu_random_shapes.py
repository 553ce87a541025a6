"""Random image sizes, row parts (count, block height), meshes, frames of the orbit and light
counts: hit ids of the whole frame equal the CPU oracle's, RGB within 1/255, and every part
equals the whole frame bit for bit (tools/fuzz_parts.py runs the same cases in larger numbers)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_against_the_oracle(seed):
    import fuzz_parts
    assert fuzz_parts.run(12, seed, verbose=False) == 0
