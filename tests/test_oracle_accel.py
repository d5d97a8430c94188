"""The oracle's own bounding-volume hierarchy (oracle.c, oracle_accel_build) must return exactly what the oracle's scan
returns -- the same primitive and the same t for every ray, ties included -- since its only purpose is to make the
oracle usable on scenes the scan is too slow for (a million triangles)."""
import numpy as np
import pytest

import scenes_util as S
from rustraytracer_amd import _abi
from test_gpu_lbvh import mesh_scene


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def _check(O, flat, n=6000, seed=3):
    o, d = S.secondary_rays(flat, O, n, seed)
    scan = O.intersect(flat, o, d)
    with O.accelerated(flat):
        tree = O.intersect(flat, o, d)
    assert _same(scan, tree)
    return (scan[0] >= 0).mean()


@pytest.mark.parametrize("name", ["cornell_box", "smallpt"])
def test_shipped_scenes(O, scenes, name):
    assert _check(O, scenes[name]) > 0.5


def test_every_primitive_type_and_bent_quads(O):
    _check(O, S.all_prims_scene())
    for flat, quads in S.bent_quad_scenes().values():
        o, d = S.bent_quad_rays(quads, seed=5)
        scan = O.intersect(flat, o, d)
        with O.accelerated(flat):
            assert _same(scan, O.intersect(flat, o, d))


def test_mesh_of_triangles_and_quads(O):
    assert _check(O, mesh_scene(3000, with_quads=False), n=4000) > 0.9


def test_ties(O):
    """Coincident primitives: the scan lets a quad or sphere replace an earlier hit at equal t and a triangle not; the
    tree must pick the same one whatever order it visits them in."""
    quad = [[-1, -1, -3], [1, -1, -3], [1, 1, -3], [-1, 1, -3]]
    tri = [[-1, -1, -3], [1, -1, -3], [1, 1, -3]]
    prims = []
    for i in range(40):  # enough primitives for several tree levels, in alternating order
        prims.append((3, 0, i, quad) if i % 3 != 1 else (2, 1, i, tri))
    prims.append((3, 2, 0, [[-5, -5, -6], [5, -5, -6], [5, 5, -6], [-5, 5, -6]]))
    mats = [(_abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)] * 3
    flat = S.make_scene(prims, mats)
    rng = np.random.default_rng(1)
    n = 3000
    o = np.zeros((n, 3))
    tgt = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-1.2, 1.2, n), np.full(n, -3.0)], axis=1)
    d = tgt / np.linalg.norm(tgt, axis=1, keepdims=True)
    scan = O.intersect(flat, o, d)
    with O.accelerated(flat):
        tree = O.intersect(flat, o, d)
    assert _same(scan, tree)
    on = scan[0] >= 0
    assert on.mean() > 0.5 and len(np.unique(scan[1][on & (scan[0] == 0)])) == 1  # the last quad of the pile wins


def test_render_is_unchanged(O, cfg, cornell):
    rc = cfg.stratified_config(24, 24, 3, 3, seed=9)
    a, sa = O.render_accumulate(cornell, rc.render_desc(), device_numbers=True)
    with O.accelerated(cornell):
        b, sb = O.render_accumulate(cornell, rc.render_desc(), device_numbers=True)
    assert np.array_equal(a, b) and sa["rays"] == sb["rays"]
    flat = mesh_scene(2000)
    rc = cfg.stratified_config(16, 16, 2, 2, seed=10)
    a, _ = O.render_accumulate(flat, rc.render_desc(flags=_abi.RENDER_NEE), device_numbers=True)
    with O.accelerated(flat):
        b, _ = O.render_accumulate(flat, rc.render_desc(flags=_abi.RENDER_NEE), device_numbers=True)
    assert np.array_equal(a, b)
