"""Hit-index parity on geometry chosen to be hard for an fp32 filter: sub-pixel triangles,
slivers, scales from 1e-3 to 20 units in one scene, a scene 600 units from the origin, rays
aimed exactly at edges and vertices, rays starting 0.001 off a surface (the shadow-ray
pattern of light_model.c:61-63) and rays grazing a triangle's plane.  Closest hit, any-hit
and t/u/v must all equal the fp64 oracle: the north star allows 0.01 % of pixels to differ,
this asserts none do."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def test_hard_geometry_has_no_mismatch():
    import stress_parity
    bad, rays, hits = stress_parity.run(seeds=1, rays=12000, verbose=False)
    assert rays == 6 * 4 * 12000 and hits > rays // 2
    assert bad == 0, f"{bad} of {rays} rays differ from the oracle"


def test_hard_geometry_frames_have_no_mismatch():
    """The same scenes through the pixel loop: pixel-grid primary sweep, the per-light shadow
    sweeps (one light 0.05 units off a triangle, one 500 units away) and shading."""
    import stress_parity
    bad_hit, bad_rgb, pixels, hits = stress_parity.run_frames(seeds=1, verbose=False)
    assert pixels == 6 * 128 * 96 and hits > 0
    assert bad_hit == 0, f"{bad_hit} of {pixels} pixels differ from the oracle in hit or t/u/v"
    assert bad_rgb == 0, f"{bad_rgb} of {pixels} pixels are off by more than 1/255"
