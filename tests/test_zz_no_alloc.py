"""rrt_intersect_static, the entry point that allocates nothing (rays and results staged in a fixed array inside the
library's device image), against rrt_intersect through the C ABI: bit-identical results on two calls in a row, the
second with a smaller input (so anything the first call left in the array would show), scan and tree, both
precisions; the size limit and the residency requirement are reported as errors."""
import numpy as np
import pytest

import scenes_util as S
from rustraytracer_amd import _abi

pytestmark = pytest.mark.gpu


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def test_static_equals_allocating_entry_point_twice(gpu, O):
    api, be = gpu
    flat = S.all_prims_scene()
    sc = api.Scene(flat)
    o_big, d_big = S.secondary_rays(flat, O, 30000, seed=21)      # 30000 camera rays + rays from their hit points
    assert 40000 < len(o_big) <= _abi.INTERSECT_STATIC_MAX_RAYS
    o_small, d_small = S.secondary_rays(flat, O, 700, seed=22)
    for prec in (api.PRECISION_F32, api.PRECISION_F64):
        for accel in (api.ACCEL_LINEAR, api.ACCEL_LBVH, api.ACCEL_SAH, api.ACCEL_SAH_GPU):
            want_big = be.intersect(sc, o_big, d_big, precision=prec, accel=accel)   # also makes scene (and tree) resident
            want_small = be.intersect(sc, o_small, d_small, precision=prec, accel=accel)
            import torch

            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info(0)[0]   # hipMemGetInfo: what the device has free, all processes
            got_big = be.intersect_static(sc, o_big, d_big, precision=prec, accel=accel)
            free_mid = torch.cuda.mem_get_info(0)[0]
            got_small = be.intersect_static(sc, o_small, d_small, precision=prec, accel=accel)
            # the entry point's defining property: no device memory is taken, during or after the calls (the GPU box
            # runs this process alone, so nothing else moves the figure)
            assert torch.cuda.mem_get_info(0)[0] == free_before == free_mid, (prec, accel)
            assert _same(got_big, want_big), (prec, accel)
            assert _same(got_small, want_small), (prec, accel)
            assert (want_big[0] >= 0).sum() > 1000 and (want_big[0] < 0).sum() > 0
    # against the oracle too (f64 ids exact), so that both entry points are not merely wrong together
    ref = O.intersect(flat, o_small, d_small)
    got = be.intersect_static(sc, o_small, d_small, precision=api.PRECISION_F64, accel=api.ACCEL_LINEAR)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_static_limits_are_errors(gpu, O):
    api, be = gpu
    flat = S.all_prims_scene()
    sc = api.Scene(flat)
    n = _abi.INTERSECT_STATIC_MAX_RAYS + 1
    o = np.zeros((n, 3))
    d = np.tile([0.0, 0.0, -1.0], (n, 1))
    with pytest.raises(api.RrtError) as e:   # the scene has never been used on the device
        be.intersect_static(sc, o[:8], d[:8])
    assert e.value.code == -1 and "resident" in str(e.value)
    be.intersect(sc, o[:8], d[:8], accel=api.ACCEL_LINEAR)
    with pytest.raises(api.RrtError) as e:   # resident, but its tree is not built
        be.intersect_static(sc, o[:8], d[:8], accel=api.ACCEL_LBVH)
    assert e.value.code == -1 and "tree is not built" in str(e.value)
    be.intersect(sc, o[:8], d[:8], accel=api.ACCEL_LBVH)
    be.intersect_static(sc, o[:8], d[:8], accel=api.ACCEL_LBVH)
    with pytest.raises(api.RrtError) as e:   # each kind of tree is built by a call that asks for that kind
        be.intersect_static(sc, o[:8], d[:8], accel=api.ACCEL_SAH)
    assert e.value.code == -1 and "tree is not built" in str(e.value)
    with pytest.raises(api.RrtError) as e:
        be.intersect_static(sc, o, d, accel=api.ACCEL_LINEAR)
    assert e.value.code == -1 and "at most" in str(e.value)
    full = be.intersect_static(sc, o[:-1], d[:-1], accel=api.ACCEL_LINEAR)   # exactly the maximum is served
    assert _same(full, be.intersect(sc, o[:-1], d[:-1], accel=api.ACCEL_LINEAR))
    assert be.intersect_static(sc, o[:0], d[:0])[0].size == 0
