"""Oracle vs the reference's shape tests (src/shapes/*.rs #[cfg(test)]) and mesh loading."""
import os

import numpy as np
import pytest

from conftest import ROOT


def _unit(v):
    return v / np.linalg.norm(v)


def _ray_through(rng, anchor, reach):
    """Two random points around `anchor` (the reference's construction) -> (origin, unit dir)."""
    p1 = anchor + _unit(rng.uniform(-1, 1, 3)) * rng.uniform(0, reach)
    p2 = anchor + _unit(rng.uniform(-1, 1, 3)) * rng.uniform(0, reach)
    return p1, _unit(p2 - p1)


def test_sphere_geometric_equals_analytic(O):  # src/shapes/sphere.rs:129-179
    rng = np.random.default_rng(10)
    hits = 0
    for _ in range(500):
        c, r = rng.uniform(-10, 10, 3), rng.uniform(0.1, 10)
        o, d = _ray_through(rng, c, 2 * r)
        prim = O.make_prim(0, [*c, r])
        g = O.prim_intersect(prim, o, d, 0.0, 100.0)
        a = O.sphere_intersect_analytic([*c, r], o, d, 0.0, 100.0)
        assert (g is None) == (a is None)
        if g is not None:
            hits += 1
            t, p, n = g
            assert abs(t - a) < 1e-6
            assert np.allclose(p, o + t * d, atol=1e-12)
            assert np.allclose(n, (p - c) / r, atol=1e-12)  # outward, never flipped
    assert hits > 100


def test_plane_hit_lies_on_plane(O):  # src/shapes/plane.rs:71-123
    rng = np.random.default_rng(11)
    hits = 0
    for _ in range(500):
        point, normal = rng.uniform(-10, 10, 3), _unit(rng.uniform(-1, 1, 3))
        o, d = _ray_through(rng, point, 10)
        h = O.prim_intersect(O.make_prim(1, [*point, *normal]), o, d, 0.0, 100.0)
        if h is not None:
            hits += 1
            t, p, n = h
            assert abs(np.dot(p - point, normal)) < 1e-6
            assert np.allclose(n, normal)
    assert hits > 50


def test_plane_parallel_reject_and_inclusive_range(O):
    prim = O.make_prim(1, [0, 0, 0, 0, 0, 1])
    assert O.prim_intersect(prim, [0, 0, 1], [1, 0, 0], 0.0, 100.0) is None  # |n.d| < 1e-6
    t = O.prim_intersect(prim, [0, 0, 2], [0, 0, -1], 0.0, 2.0)
    assert t is not None and t[0] == 2.0  # t == t_max is accepted (plane.rs:33)
    assert O.prim_intersect(prim, [0, 0, 2], [0, 0, -1], 0.0, 1.999) is None


def test_triangle_hit_on_plane_and_barycentrics(O):  # src/shapes/triangle.rs:121-185
    rng = np.random.default_rng(12)
    hits = 0
    for _ in range(1000):
        v = rng.uniform(-10, 10, (3, 3))
        o, d = _ray_through(rng, v[0], 10)
        h = O.prim_intersect(O.make_prim(2, v), o, d, 0.0, 100.0)
        if h is not None:
            hits += 1
            t, p, n = h
            normal = _unit(np.cross(v[1] - v[0], v[2] - v[0]))
            assert abs(np.dot(p - v[0], normal)) < 1e-3
            assert np.allclose(n, normal, atol=1e-12)
            denom = np.linalg.norm(np.cross(v[1] - v[0], v[2] - v[0]))
            bu = np.linalg.norm(np.cross(v[2] - v[0], p - v[0])) / denom
            bv = np.linalg.norm(np.cross(v[0] - v[1], p - v[1])) / denom
            assert 0 <= bu <= 1 + 1e-9 and 0 <= bv <= 1 + 1e-9 and bu + bv <= 1 + 1e-9
    assert hits > 50


def test_triangle_range_is_open(O):
    tri = O.make_prim(2, [[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    o, d = [0.25, 0.25, 1.0], [0, 0, -1.0]
    assert O.prim_intersect(tri, o, d, 0.0, 1.0) is None  # t == t_max rejected (triangle.rs:56)
    assert O.prim_intersect(tri, o, d, 0.0, 1.0 + 1e-9) is not None
    assert O.prim_intersect(tri, o, d, 1.0, 2.0) is None  # t == t_min rejected


def test_quad_predicates(O):  # src/shapes/quadrilateral.rs:225-266
    rng = np.random.default_rng(13)
    for _ in range(200):
        v = np.zeros((4, 3))
        v[:3, :2] = rng.uniform(-10, 10, (3, 2))
        if rng.random() < 0.5:
            v[3, :2] = rng.uniform(-10, 10, 2)
            assert O.quad_predicates(v)[0]
        else:
            v[3] = rng.uniform(-10, 10, 3)
            if abs(v[3, 2]) > 1.0:
                assert not O.quad_predicates(v)[0]
    for _ in range(200):
        base = [[-1, 1, 0], [-1, -1, 0], [1, -1, 0]]
        if rng.random() < 0.5:
            q = np.array(base + [[rng.random(), rng.random(), 0]])
            assert O.quad_predicates(q) == (True, True)
        else:
            q = np.array(base + [[-rng.random(), -rng.random(), 0]])
            cop, conv = O.quad_predicates(q)
            assert cop and not conv


def test_python_quad_predicates_match_oracle(O, cfg):
    rng = np.random.default_rng(14)
    for _ in range(300):
        q = rng.uniform(-5, 5, (4, 3))
        if rng.random() < 0.7:
            q[:, 2] = 0.0
        cop, conv = O.quad_predicates(q)
        assert bool(cfg.are_points_coplanar(*q)) == cop
        assert bool(cfg.is_quadrilateral_convex(*q)) == conv


def test_quad_equals_two_triangles(O):  # src/shapes/quadrilateral.rs:268-326
    rng = np.random.default_rng(15)
    checked = 0
    for _ in range(2000):
        v = np.array([[rng.uniform(-10, 0), rng.uniform(0, 10), 0], [rng.uniform(-10, 0), rng.uniform(-10, 0), 0],
                      [rng.uniform(0, 10), rng.uniform(-10, 0), 0], [rng.uniform(0, 10), rng.uniform(0, 10), 0]])
        if not O.quad_predicates(v)[1]:
            continue
        p1 = np.array([rng.uniform(-10, 10), rng.uniform(-10, 10), rng.uniform(-10, 0)])
        p2 = np.array([rng.uniform(-10, 10), rng.uniform(-10, 10), rng.uniform(0, 10)])
        o, d = p1, _unit(p2 - p1)
        hq = O.prim_intersect(O.make_prim(3, v), o, d, 0.0, 100.0)
        h0 = O.prim_intersect(O.make_prim(2, v[[0, 1, 2]]), o, d, 0.0, 100.0)
        h1 = O.prim_intersect(O.make_prim(2, v[[0, 2, 3]]), o, d, 0.0, 100.0)
        ht = min([h for h in (h0, h1) if h is not None], key=lambda h: h[0], default=None)
        assert (hq is None) == (ht is None)
        if hq is not None:
            assert np.linalg.norm(hq[1] - ht[1]) < 1e-6
            checked += 1
    assert checked > 200


def test_quad_normal_uses_first_three_vertices(O):
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=float)
    h = O.prim_intersect(O.make_prim(3, v), [0.9, 0.9, 1], [0, 0, -1], 0.0, 100.0)  # alpha+beta > 1 branch
    assert h is not None and np.allclose(h[2], [0, 0, 1]) and abs(h[0] - 1) < 1e-15
    assert O.prim_intersect(O.make_prim(3, v), [1.1, 0.5, 1], [0, 0, -1], 0.0, 100.0) is None


def test_load_mesh_known_answer(cfg):  # src/shapes/utils.rs:126-133
    vertices, normals, faces = cfg.load_mesh(os.path.join(ROOT, "assets", "test.ply"))
    assert len(vertices) == 24 and len(normals) == 24 and len(faces) == 6
    assert np.allclose(np.linalg.norm(normals, axis=1), 1.0)


def test_ply_sizes_match_lfs_stubs():
    # byte sizes recorded in the reference's git-LFS pointer files
    sizes = {"configs/cornell_box/Light_0000_m000.ply": 409, "configs/cornell_box/cornell_box_0000_m000.ply": 636,
             "configs/cornell_box/cornell_box_0000_m001.ply": 409, "configs/cornell_box/cornell_box_0000_m002.ply": 409,
             "configs/cornell_box/large_box_0000_m000.ply": 975, "configs/cornell_box/small_box_0000_m000.ply": 975,
             "assets/test.ply": 975, "assets/large_box_0000_m000.ply": 975}
    for rel, size in sizes.items():
        assert os.path.getsize(os.path.join(ROOT, rel)) == size, rel


def test_load_mesh_rejects(cfg, tmp_path):
    def write(name, header_props, vertex_rows, faces, fmt="ascii"):
        p = tmp_path / name
        lines = ["ply", f"format {fmt} 1.0", f"element vertex {len(vertex_rows)}"] + header_props + [
            f"element face {len(faces)}", "property list uchar uint vertex_indices", "end_header"]
        lines += [" ".join(str(x) for x in r) for r in vertex_rows]
        lines += [" ".join(str(x) for x in [len(f), *f]) for f in faces]
        p.write_text("\n".join(lines) + "\n")
        return str(p)

    fl = [f"property float {k}" for k in ("x", "y", "z", "nx", "ny", "nz")]
    sq = [[0, 0, 0, 0, 0, 1], [1, 0, 0, 0, 0, 1], [1, 1, 0, 0, 0, 1], [0, 1, 0, 0, 0, 1]]
    ok = write("ok.ply", fl, sq, [[0, 1, 2, 3], [0, 1, 2]])
    v, n, f = cfg.load_mesh(ok)
    assert len(v) == 4 and len(f) == 2
    with pytest.raises(cfg.ConfigError, match="Unsupported mesh format"):
        cfg.load_mesh("mesh.obj")
    dbl = ["property double x"] + fl[1:]
    with pytest.raises(cfg.ConfigError, match="x's type unrecognized"):
        cfg.load_mesh(write("dbl.ply", dbl, sq, [[0, 1, 2]]))
    with pytest.raises(cfg.ConfigError, match="Invalid mesh"):
        cfg.load_mesh(write("two.ply", fl, sq, [[0, 1]]))
    bent = [r[:] for r in sq]
    bent[3] = [0, 1, 0.5, 0, 0, 1]
    with pytest.raises(cfg.ConfigError, match="coplanar"):
        cfg.load_mesh(write("bent.ply", fl, bent, [[0, 1, 2, 3]]))
    dart = [r[:] for r in sq]
    dart[2] = [0.2, 0.2, 0, 0, 0, 1]
    with pytest.raises(cfg.ConfigError, match="non-convex"):
        cfg.load_mesh(write("dart.ply", fl, dart, [[0, 1, 2, 3]]))
    penta = sq + [[0.5, 1.5, 0, 0, 0, 1]]
    with pytest.raises(cfg.ConfigError, match="not supported"):
        cfg.load_mesh(write("penta.ply", fl, penta, [[0, 1, 2, 4, 3]]))


def test_ply_binary_big_endian_and_mixed_faces(cfg, tmp_path):
    import struct

    verts = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (2, 0, 0)]
    faces = [(0, 1, 2, 3), (1, 4, 2)]
    for fmt, e in (("binary_big_endian", ">"), ("binary_little_endian", "<")):
        p = tmp_path / f"{fmt}.ply"
        hdr = ("ply\nformat %s 1.0\nelement vertex 5\n" % fmt + "".join(
            f"property float {k}\n" for k in ("x", "y", "z", "nx", "ny", "nz"))
            + "element face 2\nproperty list uchar uint vertex_indices\nend_header\n")
        with open(p, "wb") as f:
            f.write(hdr.encode())
            for v in verts:
                f.write(struct.pack(e + "6f", *v, 0, 0, 1))
            for fc in faces:
                f.write(struct.pack(e + "B%dI" % len(fc), len(fc), *fc))
        v, n, fs = cfg.load_mesh(str(p))
        assert np.allclose(v, verts) and [list(x) for x in fs] == [list(x) for x in faces]
