"""The two trees (the radix tree built on the GPU, the surface-area tree built on the host): either must return the
same closest hit as the linear scan (ids AND distance, bit for bit, in both precisions), including the reference's
tie rule, with unbounded planes kept beside the tree."""
import os
import sys

import numpy as np
import pytest

import scenes_util as S
from conftest import ROOT
from rustraytracer_amd import _abi

sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_mesh  # noqa: E402

pytestmark = pytest.mark.gpu

TREES = (_abi.ACCEL_LBVH, _abi.ACCEL_SAH, _abi.ACCEL_SAH_GPU)


def mesh_scene(n_tri, seed=2, with_quads=False):
    """A displaced sphere of triangles (one mesh object) in front of the camera, inside an emissive sphere,
    above a Lambertian plane; optionally a second mesh object made of quads."""
    v, _, f = make_mesh.displaced_sphere(n_tri, radius=1.2, bump=0.2, seed=seed)
    tri = v[f].astype(np.float64) + np.array([0, 0, -4.0])
    p = np.zeros(len(f) + 2 + (6 if with_quads else 0), dtype=_abi.PRIM_DTYPE)
    p["type"][: len(f)] = _abi.PRIM_TRIANGLE
    p["prim_id"][: len(f)] = np.arange(len(f))
    p["v"][: len(f), :9] = tri.reshape(len(f), 9)
    k = len(f)
    p["type"][k], p["object_id"][k] = _abi.PRIM_SPHERE, 1
    p["v"][k, :4] = [0, 0, 0, 40.0]
    p["type"][k + 1], p["object_id"][k + 1] = _abi.PRIM_PLANE, 2
    p["v"][k + 1, :6] = [0, -1.5, 0, 0, 1, 0]
    mats = [(_abi.MAT_LAMBERTIAN, (0.7, 0.7, 0.7), 0.0), (_abi.MAT_EMISSIVE, (1, 1, 1), 0.0),
            (_abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)]
    m = np.zeros(len(mats) + (1 if with_quads else 0), dtype=_abi.MATERIAL_DTYPE)
    for i, (typ, rgb, param) in enumerate(mats):
        m["type"][i], m["rgb"][i], m["param"][i] = typ, rgb, param
    from rustraytracer_amd import config
    cam = config.perspective_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 50.0, 1.0)
    return config.FlatScene(cam, p, m)


def cornell_with_cube(cfg, cornell):
    """The cornell_cube benchmark scene: the Cornell box plus assets/test.ply (a unit cube of 6 quads) as a 7th object,
    scaled to 0.12 m and set on the floor in front of the boxes, Phong material."""
    import copy
    verts, _, faces = cfg.load_mesh(os.path.join(ROOT, "assets", "test.ply"))
    m = np.diag([0.12, 0.12, 0.12, 1.0])
    m[:3, 3] = [-0.40, 0.12, 0.06]
    verts = cfg.transform_point3(m, verts)
    extra = np.zeros(len(faces), dtype=_abi.PRIM_DTYPE)
    extra["type"] = _abi.PRIM_QUAD
    extra["object_id"] = cornell.n_objects
    extra["prim_id"] = np.arange(len(faces))
    extra["v"][:, :12] = verts[np.asarray(faces)].reshape(len(faces), 12)
    mat = np.zeros(1, dtype=_abi.MATERIAL_DTYPE)
    mat["type"], mat["rgb"], mat["param"] = _abi.MAT_PHONG, (0.8, 0.8, 0.3), 30.0
    return cfg.FlatScene(copy.deepcopy(cornell.camera), np.concatenate([cornell.prims, extra]),
                         np.concatenate([cornell.materials, mat]))


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("name", ["cornell_box", "smallpt", "all_prims", "cornell_cube", "mesh3k"])
def test_lbvh_equals_linear_scan(gpu, O, cfg, scenes, cornell, name):
    api, be = gpu
    flat = {"all_prims": S.all_prims_scene, "cornell_cube": lambda: cornell_with_cube(cfg, cornell),
            "mesh3k": lambda: mesh_scene(3000)}.get(name, lambda: scenes[name])()
    sc = api.Scene(flat)
    o, d = S.secondary_rays(flat, O, 40000, seed=23)
    ref = O.intersect(flat, o, d)
    for prec in (api.PRECISION_F64, api.PRECISION_F32):
        lin = be.intersect(sc, o, d, precision=prec, accel=api.ACCEL_LINEAR)
        for tree in TREES:
            bvh = be.intersect(sc, o, d, precision=prec, accel=tree)
            assert _same(lin, bvh), f"{name} prec {prec} tree {tree}: tree and scan disagree"
    bvh64 = be.intersect(sc, o, d, precision=api.PRECISION_F64, accel=api.ACCEL_LBVH)
    bad = (bvh64[0] != ref[0]) | (bvh64[1] != ref[1])
    with np.errstate(invalid="ignore"):
        tie = bad & (np.abs(bvh64[2] - ref[2]) <= 1e-12 * np.abs(ref[2]))  # coincident surfaces
    assert not np.any(bad & ~tie)


def test_lbvh_tie_rule(gpu):
    api, be = gpu
    quad = [[-1, -1, -2], [1, -1, -2], [1, 1, -2], [-1, 1, -2]]
    tri = [[-1, -1, -2], [3, -1, -2], [-1, 3, -2]]
    mats = [(_abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)] * 4
    o, d = [[0.0, 0.0, 0.0]], [[0.0, 0.0, -1.0]]
    cases = [([(3, 0, 0, quad), (3, 1, 0, quad)], 1), ([(2, 0, 0, tri), (2, 1, 0, tri)], 0),
             ([(3, 0, 0, quad), (2, 1, 0, tri)], 0), ([(2, 0, 0, tri), (3, 1, 0, quad)], 1),
             ([(2, 0, 0, tri), (3, 1, 0, quad), (2, 2, 0, tri), (1, 3, 0, [0, 0, -2, 0, 0, 1])], 3),
             ([(1, 3, 0, [0, 0, -2, 0, 0, 1]), (2, 0, 0, tri), (3, 1, 0, quad), (2, 2, 0, tri)], 1),
             ([(2, 0, 0, tri), (2, 1, 0, tri), (2, 2, 0, tri), (3, 3, 0, quad), (2, 0, 1, tri)], 3)]
    for prims, winner in cases:
        sc = api.Scene(S.make_scene(prims, mats))
        for prec in (api.PRECISION_F32, api.PRECISION_F64):
            for accel in (api.ACCEL_LINEAR,) + TREES:
                assert be.intersect(sc, o, d, precision=prec, accel=accel)[0][0] == winner, (prims, accel)


def test_lbvh_large_mesh_primary_parity(gpu, O):
    """60 k triangles: the oracle's linear scan is still affordable for a few thousand rays."""
    api, be = gpu
    flat = mesh_scene(60000, seed=5)
    sc = api.Scene(flat)
    rng = np.random.default_rng(4)
    u, v = rng.random(3000), rng.random(3000)
    ref = O.trace_primary(flat, u, v)
    got = be.trace_primary(sc, u, v, precision=api.PRECISION_F64)  # AUTO -> tree for this size
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    hit = ref[0] >= 0
    assert np.all(np.abs(got[2][hit] - ref[2][hit]) <= 1e-9 * ref[2][hit])
    assert (ref[0] == 0).sum() > 500  # the mesh fills a good part of the view
    got32 = be.trace_primary(sc, u, v, precision=api.PRECISION_F32)
    bad = (got32[0] != ref[0]) | (got32[1] != ref[1])
    assert bad.sum() == 0  # rays through a shared edge of two facets within f32 resolution are decided in f64
    ok = ~bad & hit
    assert np.max(np.abs(got32[2][ok] - ref[2][ok]) / ref[2][ok]) <= 1e-5
    lin = be.trace_primary(sc, u, v, precision=api.PRECISION_F32, accel=api.ACCEL_LINEAR)
    assert _same(lin, got32)


def test_render_with_lbvh_matches_linear(gpu, cfg, cornell):
    """Same seed, same scene: every path meets the same surfaces through either structure, so the two
    renders agree to float summation order."""
    api, be = gpu
    flat = cornell_with_cube(cfg, cornell)
    sc = api.Scene(flat)
    rc = cfg.stratified_config(48, 48, 8, 8, seed=2)
    a, sa = be.render_accumulate(sc, rc.render_desc(accel=api.ACCEL_LINEAR))
    for tree in TREES:
        b, sb = be.render_accumulate(sc, rc.render_desc(accel=tree))
        assert sa["rays"] == sb["rays"] and sa["bounces"] == sb["bounces"]
        assert np.allclose(a, b, rtol=1e-5, atol=1e-5)
        assert sb["node_tests"] > 0 and sb["prim_tests"] < sa["prim_tests"]
    assert sa["bounces"][2] > 0  # the Phong cube is hit


def test_degenerate_trees(gpu):
    """0, 1 and 2 bounded primitives, and a scene of planes only."""
    api, be = gpu
    mats = [(_abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)] * 3
    tri = [[-1, -1, -2], [3, -1, -2], [-1, 3, -2]]
    o, d = [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]], [[0.0, 0.0, -1.0], [0.0, 1.0, 0.0]]
    for prims, expect in (([(1, 0, 0, [0, 0, -3, 0, 0, 1])], [0, -1]),
                          ([(2, 1, 0, tri)], [1, -1]),
                          ([(2, 1, 0, tri), (0, 2, 0, [0, 5, 0, 1.0])], [1, 2]),
                          ([(1, 0, 0, [0, 0, -3, 0, 0, 1]), (2, 1, 0, tri), (0, 2, 0, [0, 5, 0, 1.0])], [1, 2])):
        sc = api.Scene(S.make_scene(prims, mats))
        for prec in (api.PRECISION_F32, api.PRECISION_F64):
            for tree in TREES:
                got = be.intersect(sc, o, d, precision=prec, accel=tree)
                assert list(got[0]) == expect, (prims, got, tree)


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.5, 0.5, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: n // 8, 0] = 0.0  # a share of rays with an exactly-zero direction component (infinite slabs)
    d[: n // 16, 1] = 0.0
    d[: n // 8] /= np.linalg.norm(d[: n // 8], axis=1, keepdims=True)
    return o, d


@pytest.mark.parametrize("case", ["one", "two", "three", "planes_only", "identical", "degenerate", "far_offset"])
def test_lbvh_edge_cases(gpu, case):
    """Tiny trees, no tree at all, coincident and zero-area primitives, a scene far from the origin:
    tree and scan must agree bit for bit, axis-parallel rays included."""
    api, be = gpu
    tri = lambda z, dx=0.0: [[-1 + dx, -1, z], [2 + dx, -1, z], [-1 + dx, 2, z]]
    quad = lambda z: [[-1, -1, z], [1, -1, z], [1, 1, z], [-1, 1, z]]
    mats = [(_abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)]
    shift = np.zeros(3)
    if case == "one":
        prims = [(2, 0, 0, tri(-2))]
    elif case == "two":
        prims = [(2, 0, 0, tri(-2)), (0, 0, 1, [0, 0, 3, 1.0])]
    elif case == "three":
        prims = [(2, 0, 0, tri(-2)), (3, 0, 1, quad(-3)), (0, 0, 2, [0.5, 0.5, 2.5, 0.75])]
    elif case == "planes_only":
        prims = [(1, 0, 0, [0, 0, -2, 0, 0, 1]), (1, 0, 1, [0, -2, 0, 0, 1, 0]), (1, 0, 2, [3, 0, 0, 1, 0, 0])]
    elif case == "identical":  # equal Morton codes: the position in the sorted order breaks the tie
        prims = [(3 if i % 3 else 2, 0, i, quad(-2) if i % 3 else tri(-2)) for i in range(40)]
    elif case == "degenerate":  # zero-area faces never hit anything (|n.d| < 1e-6) but must not break the build
        prims = [(2, 0, 0, [[0, 0, -2]] * 3), (2, 0, 1, [[0, 0, -2], [1, 0, -2], [2, 0, -2]]), (2, 0, 2, tri(-3)),
                 (3, 0, 3, quad(-4)), (2, 0, 4, [[5, 5, 5]] * 3)]
    else:  # far_offset: coordinates ~1e4 with unit-size geometry
        shift = np.array([1.0e4, -2.0e4, 3.0e4])
        prims = [(2, 0, i, (np.array(tri(-2 - 0.1 * i, 0.05 * i)) + shift)) for i in range(30)]
        prims.append((0, 0, 30, [*(np.array([0.0, 0.0, 2.0]) + shift), 0.5]))
    sc = api.Scene(S.make_scene(prims, mats))
    o, d = _rays(20000, seed=11)
    o = o + shift
    for prec in (api.PRECISION_F64, api.PRECISION_F32):
        lin = be.intersect(sc, o, d, precision=prec, accel=api.ACCEL_LINEAR)
        for tree in TREES:
            bvh = be.intersect(sc, o, d, precision=prec, accel=tree)
            assert _same(lin, bvh), f"{case} prec {prec} tree {tree}"
        if case != "degenerate":
            assert (lin[0] >= 0).sum() > 1000
    # and a render through both pipelines on the tree
    from rustraytracer_amd import config as cfg
    rc = cfg.stratified_config(32, 32, 2, 2)
    c = be.render_accumulate(sc, rc.render_desc(accel=api.ACCEL_LINEAR, pipeline=api.PIPELINE_FUSED))
    for tree in TREES:
        a = be.render_accumulate(sc, rc.render_desc(accel=tree, pipeline=api.PIPELINE_FUSED))
        b = be.render_accumulate(sc, rc.render_desc(accel=tree, pipeline=api.PIPELINE_WAVEFRONT))
        assert a[1]["rays"] == b[1]["rays"] == c[1]["rays"] and a[1]["bounces"] == b[1]["bounces"] == c[1]["bounces"]


def test_boxes_growing_along_a_line(gpu):
    """Triangles whose size and spacing grow geometrically along x: every cluster of the GPU builder prefers its left
    neighbour, so only one pair per round would merge by cost alone (the builder then pairs by position); both trees
    must still be built quickly and agree with the scan."""
    import time
    api, be = gpu
    n = 20000
    x = 1.0005 ** np.arange(n)  # 1 .. 22000
    s = 2e-4 * x
    prims = []
    tri = np.zeros((n, 3, 3))
    tri[:, 0] = np.stack([x, -s, -s - 5], axis=1)
    tri[:, 1] = np.stack([x + s, s, -5 + 0 * x], axis=1)
    tri[:, 2] = np.stack([x - s, s, s - 5], axis=1)
    p = np.zeros(n, dtype=_abi.PRIM_DTYPE)
    p["type"] = _abi.PRIM_TRIANGLE
    p["prim_id"] = np.arange(n)
    p["v"][:, :9] = tri.reshape(n, 9)
    m = np.zeros(1, dtype=_abi.MATERIAL_DTYPE)
    m["type"], m["rgb"] = _abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5)
    from rustraytracer_amd import config
    flat = config.FlatScene(config.perspective_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 50.0, 1.0), p, m)
    sc = api.Scene(flat)
    rng = np.random.default_rng(5)
    k = rng.integers(0, n, 4000)
    target = tri[k].mean(axis=1) + rng.normal(size=(4000, 3)) * s[k, None] * 0.5
    o = target + np.array([0.0, 0.0, 3.0]) + rng.normal(size=(4000, 3)) * 0.3
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    lin = be.intersect(sc, o, d, accel=api.ACCEL_LINEAR)
    assert (lin[0] >= 0).sum() > 500
    for tree in TREES:
        t0 = time.time()
        got = be.intersect(sc, o, d, accel=tree)
        assert time.time() - t0 < 10.0, tree
        assert _same(lin, got), tree


def test_empty_scene(gpu, cfg):
    api, be = gpu
    sc = api.Scene(S.make_scene([], []))
    o, d = _rays(100, seed=1)
    for accel in (api.ACCEL_LINEAR,) + TREES:
        got = be.intersect(sc, o, d, accel=accel)
        assert np.all(got[0] == -1) and np.all(np.isinf(got[2]))
    rc = cfg.stratified_config(16, 16, 2, 2)
    for pipe in (api.PIPELINE_FUSED, api.PIPELINE_WAVEFRONT):
        rgb, st = be.render_accumulate(sc, rc.render_desc(pipeline=pipe))
        assert st["samples"] == 16 * 16 * 4 and st["rays"] == st["samples"] and not rgb.any()


def test_million_triangle_scene_tree_equals_scan(gpu, cfg, O):
    """configs/mesh_1m (two generated 500k-triangle meshes in the Cornell shell, millimetre units): the
    tree must return what a linear scan over all 998 290 primitives returns, bit for bit, for primary and
    for scattered rays; the oracle's scan agrees on a smaller set (it tests 1 M triangles per ray)."""
    import make_c5_assets
    api, be = gpu
    make_c5_assets.main()
    flat = cfg.load_scene(os.path.join(ROOT, "configs", "mesh_1m", "scene.toml"))
    assert flat.n_prims == 998290
    sc = api.Scene(flat)
    rng = np.random.default_rng(9)
    u, v = rng.random(6000), rng.random(6000)
    tree = be.trace_primary(sc, u, v, precision=api.PRECISION_F32)  # AUTO -> the surface-area tree
    scan = be.trace_primary(sc, u, v, precision=api.PRECISION_F32, accel=api.ACCEL_LINEAR)
    assert _same(tree, scan)
    assert _same(be.trace_primary(sc, u, v, precision=api.PRECISION_F32, accel=api.ACCEL_LBVH), scan)
    assert (tree[0] >= 4).sum() > 1000  # a good share of the rays meets the two meshes
    # scattered rays: leave the primary hit points in random directions
    hit = tree[0] >= 0
    d0 = np.stack([flat.camera["lower_left_corner"] + uu * flat.camera["horizontal"] + vv * flat.camera["vertical"]
                   - flat.camera["origin"] for uu, vv in zip(u[hit], v[hit])])
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    o2 = flat.camera["origin"] + d0 * tree[2][hit][:, None]
    d2 = rng.normal(size=o2.shape)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    for prec in (api.PRECISION_F32, api.PRECISION_F64):
        b = be.intersect(sc, o2, d2, precision=prec, accel=api.ACCEL_LINEAR)
        for accel in (api.ACCEL_AUTO,) + TREES:
            assert _same(be.intersect(sc, o2, d2, precision=prec, accel=accel), b), (prec, accel)
    n = 200
    ref = O.trace_primary(flat, u[:n], v[:n])
    got = be.trace_primary(sc, u[:n], v[:n], precision=api.PRECISION_F64)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    h = ref[0] >= 0
    assert np.all(np.abs(got[2][h] - ref[2][h]) <= 1e-9 * ref[2][h])


@pytest.mark.parametrize("name", ["mesh3k", "mesh60k", "mesh200k_quads", "cornell_cube", "all_prims"])
def test_gpu_built_surface_area_tree_is_the_host_builders_tree(gpu, cfg, cornell, name):
    """RRT_ACCEL_SAH_GPU (what AUTO builds) makes its cuts with the host builder's rules -- same cut coordinates, bins,
    cost, tie order, the exact sweep for ranges of up to 32 primitives -- so it must arrive at the same tree: a render
    walks the same number of nodes and tests the same number of primitives through either, ray for ray, and the
    trees are equally deep.  (The node numbering differs, which no count sees.)"""
    api, be = gpu
    flat = {"mesh3k": lambda: mesh_scene(3000), "mesh60k": lambda: mesh_scene(60000, seed=5),
            "mesh200k_quads": lambda: mesh_scene(200000, seed=9, with_quads=False),
            "cornell_cube": lambda: cornell_with_cube(cfg, cornell), "all_prims": S.all_prims_scene}[name]()
    sc = api.Scene(flat)
    rc = cfg.stratified_config(96, 96, 2, 2, seed=61)
    res = {}
    for accel in (api.ACCEL_SAH, api.ACCEL_SAH_GPU):
        img, st = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F64, accel=accel, pipeline=api.PIPELINE_FUSED))
        assert st["accel_used"] == accel and st["tree_built"] == 1 and st["stack_overflows"] == 0
        res[accel] = (img, st)
    host, dev = res[api.ACCEL_SAH][1], res[api.ACCEL_SAH_GPU][1]
    print(name, "node tests", host["node_tests"], dev["node_tests"], "prim tests", host["prim_tests"], dev["prim_tests"],
          "depth", host["tree_depth"], dev["tree_depth"], "build ms", host["tree_build_ms"], dev["tree_build_ms"])
    assert host["rays"] == dev["rays"] and host["node_tests"] > 0
    assert dev["node_tests"] == host["node_tests"] and dev["prim_tests"] == host["prim_tests"]
    assert dev["tree_depth"] == host["tree_depth"]
    assert np.allclose(res[api.ACCEL_SAH][0], res[api.ACCEL_SAH_GPU][0], rtol=1e-6, atol=1e-9)
    # AUTO takes the GPU builder (for scenes large enough for a tree at all)
    _, st = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F32))
    if flat.n_prims > 72:
        assert st["accel_used"] == api.ACCEL_SAH_GPU and st["tree_built"] == 0


def test_a_tree_too_deep_for_the_stacks_is_rebuilt_with_median_cuts(cfg):
    """The traversal stacks hold three links per level of up to 48 levels; the library measures the depth of every tree
    it builds and rebuilds a deeper one with median cuts (depth <= log2 n).  RRT_DEBUG_MAX_TREE_DEPTH lowers the limit
    so that an ordinary mesh takes that path: still the scan's hits, bit for bit, a shallow tree, nothing dropped.
    (A process of its own: the limit is read once per process.)"""
    import subprocess
    code = r'''
import os, sys
sys.path.insert(0, os.environ["RRT_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RRT_ROOT"], "tests"))
import numpy as np
from oracle import oracle_py as O
from rustraytracer_amd import api, config
import scenes_util as S
import test_gpu_lbvh as T
flat = T.mesh_scene(3000)
be, sc = api.Backend(0), api.Scene(flat)
o, d = S.secondary_rays(flat, O, 20000, seed=5)
lin = be.intersect(sc, o, d, accel=api.ACCEL_LINEAR)
rc = config.stratified_config(64, 64, 2, 2)
for accel in (api.ACCEL_SAH_GPU, api.ACCEL_SAH, api.ACCEL_LBVH):
    got = be.intersect(sc, o, d, accel=accel)
    assert all(np.array_equal(a, b) for a, b in zip(lin, got)), accel
    _, st = be.render_accumulate(sc, rc.render_desc(accel=accel))
    assert st["tree_depth"] <= 12 and st["stack_overflows"] == 0, st   # ceil(log2 3000) = 12 binary levels at most
    print("accel", accel, "depth", st["tree_depth"])
print("fallback ok")
'''
    env = dict(os.environ, RRT_ROOT=ROOT, RRT_DEBUG_MAX_TREE_DEPTH="3")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0 and "fallback ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_the_host_builder_stands_in_when_the_gpu_builder_gives_up(cfg):
    """The level-by-level GPU build stops at a tree thousands of levels deep; the library then has the host builder make
    the same tree.  RRT_DEBUG_GSAH_FAIL sends an ordinary mesh down that path: the scan's hits, the host builder's depth."""
    import subprocess
    code = r'''
import os, sys
sys.path.insert(0, os.environ["RRT_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RRT_ROOT"], "tests"))
import numpy as np
from oracle import oracle_py as O
from rustraytracer_amd import api, config
import scenes_util as S
import test_gpu_lbvh as T
flat = T.mesh_scene(3000)
be, sc = api.Backend(0), api.Scene(flat)
o, d = S.secondary_rays(flat, O, 20000, seed=6)
lin = be.intersect(sc, o, d, accel=api.ACCEL_LINEAR)
rc = config.stratified_config(64, 64, 2, 2)
depth = {}
for accel in (api.ACCEL_SAH_GPU, api.ACCEL_SAH):
    got = be.intersect(sc, o, d, accel=accel)
    assert all(np.array_equal(a, b) for a, b in zip(lin, got)), accel
    _, st = be.render_accumulate(sc, rc.render_desc(accel=accel))
    depth[accel] = (st["tree_depth"], st["node_tests"], st["prim_tests"])
assert depth[api.ACCEL_SAH_GPU] == depth[api.ACCEL_SAH], depth
print("stand-in ok", depth)
'''
    env = dict(os.environ, RRT_ROOT=ROOT, RRT_DEBUG_GSAH_FAIL="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0 and "stand-in ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_two_host_threads_share_one_scene(gpu, O):
    """Two host threads, each with a backend of its own, use ONE scene at once and ask for different trees on its first
    use: the device tables and each tree are built once, by whoever comes first, and both get the scan's hits."""
    import threading
    api, be = gpu
    flat = mesh_scene(20000, seed=9)
    o, d = S.secondary_rays(flat, O, 20000, seed=8)
    lin = be.intersect(api.Scene(flat), o, d, accel=api.ACCEL_LINEAR)
    for _ in range(3):
        sc = api.Scene(flat)  # fresh: nothing resident, nothing built
        out, err = {}, []

        def work(name, accel):
            try:
                mine = api.Backend(0)
                out[name] = mine.intersect(sc, o, d, accel=accel)
            except Exception as e:  # noqa: BLE001 -- reported by the assertion below
                err.append((name, e))

        threads = [threading.Thread(target=work, args=(n, a)) for n, a in
                   (("a", api.ACCEL_SAH_GPU), ("b", api.ACCEL_LBVH), ("c", api.ACCEL_SAH_GPU), ("d", api.ACCEL_SAH))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not err, err
        for name in "abcd":
            assert _same(lin, out[name]), name
