"""Closest hit on quads that are not planar, through the C ABI: the device must intersect the second triangle's own
plane as the reference does (src/shapes/quadrilateral.rs:90-111), in the scan and beside the tree.  f64: ids exact on
every ray; f32: ids exact on every ray whose decisions are not within f32 rounding of a boundary."""
import numpy as np
import pytest

import np_reference as NP
import scenes_util as S

pytestmark = pytest.mark.gpu


def _margin(flat, quads, o, d):
    m = np.full(len(o), np.inf)
    for q in quads:
        m = np.minimum(m, NP.quad_margin(q.reshape(-1), o, d))
    return m


@pytest.mark.parametrize("name", ["mesh_bent", "free_bent"])
def test_bent_quads_against_the_oracle(gpu, O, name):
    api, be = gpu
    flat, quads = S.bent_quad_scenes()[name]
    o, d = S.bent_quad_rays(quads, seed=11)
    ref = O.intersect(flat, o, d)
    sc = api.Scene(flat)
    margin = _margin(flat, quads, o, d)
    hit = ref[0] >= 0
    assert hit.sum() > 0.3 * len(o) and (~hit).sum() > 50
    for accel in (api.ACCEL_LINEAR, api.ACCEL_LBVH, api.ACCEL_SAH, api.ACCEL_SAH_GPU):
        oid, pid, t = be.intersect(sc, o, d, precision=api.PRECISION_F64, accel=accel)
        assert np.array_equal(oid, ref[0]) and np.array_equal(pid, ref[1]), (name, accel)
        assert np.all(np.abs(t[hit] - ref[2][hit]) <= 1e-11 * np.abs(ref[2][hit]))
        oid, pid, t = be.intersect(sc, o, d, precision=api.PRECISION_F32, accel=accel)
        clear = margin > 1e-4
        assert clear.sum() > 0.8 * len(o)
        assert np.array_equal(oid[clear], ref[0][clear]) and np.array_equal(pid[clear], ref[1][clear]), (name, accel)
        both = clear & hit
        # 1e-5 relative for rays that meet the surface at a fair angle; a ray that grazes it (down to 2 degrees here)
        # divides the f32 rounding of n.o (coordinates of size |o|) by a small n.d: every surface here faces +-z
        tol = 1e-5 * np.abs(ref[2]) + 5e-7 * np.linalg.norm(o, axis=1) / np.abs(d[:, 2])
        assert np.all(np.abs(t[both] - ref[2][both]) <= tol[both])
        steep = both & (np.abs(d[:, 2]) > 0.3)
        # (an origin a few units away fixes the absolute error at a few f32 ulps of |o|, however short the ray)
        assert steep.sum() > 100
        assert np.all(np.abs(t[steep] - ref[2][steep]) <= 1e-5 * np.abs(ref[2][steep]) + 5e-7 * np.linalg.norm(o[steep], axis=1))
        assert (oid != ref[0]).sum() <= 0.02 * len(o)  # edge grazes only


def test_scan_and_tree_agree_bit_for_bit_on_bent_quads(gpu):
    api, be = gpu
    for name, (flat, quads) in S.bent_quad_scenes().items():
        o, d = S.bent_quad_rays(quads, seed=12)
        sc = api.Scene(flat)
        for prec in (api.PRECISION_F64, api.PRECISION_F32):
            a = be.intersect(sc, o, d, precision=prec, accel=api.ACCEL_LINEAR)
            for tree in (api.ACCEL_LBVH, api.ACCEL_SAH, api.ACCEL_SAH_GPU):
                b = be.intersect(sc, o, d, precision=prec, accel=tree)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (name, prec, tree)


def test_bent_quad_render_follows_the_oracle(gpu, O, cfg):
    """The bent quads inside a render: same paths as the oracle on the device's numbers (f64 kernels)."""
    api, be = gpu
    flat, _ = S.bent_quad_scenes()["free_bent"]
    # light the scene: an enclosing emissive sphere as a fourth object
    prims = np.concatenate([flat.prims, O.make_prim(0, [0, 0, 0, 30.0], object_id=3)])
    mats = np.concatenate([flat.materials, np.array([(0, 0, (1.0, 0.9, 0.8), 0.0)], dtype=flat.materials.dtype)])
    lit = cfg.FlatScene(flat.camera, prims, mats)
    rc = cfg.stratified_config(48, 48, 4, 4, seed=3)
    sc = api.Scene(lit)
    rgb, st = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F64))
    ref, rst = O.render_accumulate(lit, rc.render_desc(), device_numbers=True)
    off = np.any(np.abs(rgb - ref) > 1e-5 * np.abs(ref) + 1e-6, axis=-1)
    assert off.sum() <= 2, int(off.sum())
    assert abs(st["rays"] - rst["rays_live"]) <= 2e-4 * rst["rays_live"]
