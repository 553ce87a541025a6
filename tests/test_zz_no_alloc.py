"""rt_trace_rays_static, the entry point that allocates nothing: its outputs must equal those of
rt_trace_rays (and the oracle's) bit for bit, on two calls in a row, the second with a smaller
input, so that what the first call left in the shared static workspace is seen to be harmless.
Called through the C ABI (ctypes)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import _lib, kernel, scenes

pytestmark = pytest.mark.gpu


def _rays(rng, centre, n):
    S = centre + rng.uniform(-12, 12, size=(n, 3))
    E = centre + rng.uniform(-6, 6, size=(n, 3))
    return S, E


def test_static_entry_matches_allocating_entry_twice():
    import torch

    mesh = scenes.triangle_soup(3000, seed=21)
    fr = kernel.driver_frame(0)
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
    osc = O.Scene(mesh, X, Y, Z)
    centre = np.array([X[1:].mean(), Y[1:].mean(), Z[1:].mean()])
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    rng = np.random.default_rng(77)
    # the second call is smaller: rays 1500.. of the first call's workspace keep stale contents
    for n in (6000, 1500):
        S, E = _rays(rng, centre, n)
        # a few unused slots (NaN origin) and a zero-length ray, as the allocating path accepts
        S[5] = np.nan
        E[9] = S[9]
        hit_s, tuv_s, pt_s = sc.trace_rays(S, E, static=True)
        hit_a, tuv_a, pt_a = sc.trace_rays(S, E)
        assert torch.equal(hit_s, hit_a)
        m = hit_a >= 0
        assert int(m.sum()) > n // 20, "rays miss the mesh; test is vacuous"
        assert torch.equal(tuv_s[m], tuv_a[m]) and torch.equal(pt_s[m], pt_a[m])
        live = ~np.isnan(S[:, 0])
        ref_hit, ref_tuv, ref_pt = O.trace_rays(osc, np.nan_to_num(S), E)
        assert np.array_equal(hit_s.cpu().numpy()[live], ref_hit[live])
        mm = live & (ref_hit >= 0)
        assert np.array_equal(tuv_s.cpu().numpy()[mm], ref_tuv[mm])
        assert np.array_equal(pt_s.cpu().numpy()[mm], ref_pt[mm])
    sc.close()


def test_static_entry_after_rt_init_again():
    """The static workspace is looked up by the scene's own device, not by whichever device the
    most recent rt_init named: the same results after rt_init(0) has run once more."""
    import torch

    mesh = scenes.triangle_soup(1200, seed=4)
    fr = kernel.driver_frame(2)
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
    centre = np.array([X[1:].mean(), Y[1:].mean(), Z[1:].mean()])
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    S, E = _rays(np.random.default_rng(5), centre, 2000)
    before = sc.trace_rays(S, E, static=True)
    _lib.check(_lib.load().rt_init(0), "rt_init")
    after = sc.trace_rays(S, E, static=True)
    plain = sc.trace_rays(S, E)
    m = plain[0] >= 0
    assert int(m.sum()) > 100
    for a, b, c in zip(before, after, plain):
        assert torch.equal(a[m], b[m]) and torch.equal(b[m], c[m])
    assert torch.equal(before[0], plain[0])
    sc.close()


def test_static_entry_rejects_oversized_batches():
    import torch

    mesh = scenes.torus(300)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(kernel.driver_frame(0), "obmat"))
    lib = _lib.load()
    n = 65536 + 1
    S = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    hit = torch.empty((n,), dtype=torch.int32, device="cuda")
    rc = lib.rt_trace_rays_static(sc.h, n, S.data_ptr(), S.data_ptr(), hit.data_ptr(), None, None)
    assert rc == -6 and b"exceed" in lib.rt_last_error()      # RT_E_LIMIT
    # the largest allowed batch runs
    n = 65536
    rc = lib.rt_trace_rays_static(sc.h, n, S.data_ptr(), S.data_ptr(), hit.data_ptr(), None, None)
    assert rc == 0 and int((hit[:n] >= 0).sum()) == 0      # zero-length rays hit nothing
    sc.close()
