"""Quads that are not planar (src/shapes/quadrilateral.rs:90-111): the oracle against the independent numpy
restatement, and proof that the rays used here tell the reference's second-plane test from a same-plane shortcut."""
import numpy as np

import np_reference as NP
import scenes_util as S


def test_mesh_lift_is_what_the_loader_admits(cfg):
    q = S.bent_quad(0, 0, S.BENT_LIFT_MESH)
    assert cfg.are_points_coplanar(*q) and cfg.is_quadrilateral_convex(*q)
    assert not cfg.are_points_coplanar(*S.bent_quad(0, 0, 0.04))  # just past the 1e-3 gate


def test_oracle_equals_numpy_on_bent_quads(O):
    for name, (flat, quads) in S.bent_quad_scenes().items():
        o, d = S.bent_quad_rays(quads, seed=3)
        oid, pid, t = O.intersect(flat, o, d)
        n_oid, n_pid, n_t = NP.scene_intersect(flat.prims, o, d)
        assert np.array_equal(oid, n_oid) and np.array_equal(pid, n_pid), name
        hit = oid >= 0
        assert hit.sum() > 0.3 * len(o) and (~hit).sum() > 0
        assert np.all(np.abs(t[hit] - n_t[hit]) <= 1e-12 * np.abs(n_t[hit]))


def _same_plane_shortcut(v, o, d):
    """The quad test with the second corner judged at the FIRST plane's hit point (what a planar quad allows)."""
    v0, v1, v2, v3 = v
    n = np.cross(v1 - v0, v3 - v0)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ((v0 - o) @ n) / (d @ n)
    P = o + t[:, None] * d

    def coef(c, ea, eb, nn):
        k = np.cross(eb, nn)
        return ((P - c) @ k) / (ea @ k)
    a, b = coef(v0, v1 - v0, v3 - v0, n), coef(v0, v3 - v0, v1 - v0, n)
    n2 = np.cross(v3 - v2, v1 - v2)
    a2, b2 = coef(v2, v3 - v2, v1 - v2, n2), coef(v2, v1 - v2, v3 - v2, n2)
    return (a >= 0) & (b >= 0) & ((a + b <= 1) | ((a2 >= 0) & (b2 >= 0))) & (t >= NP.T_MIN) & (np.abs(d @ n) >= 1e-6)


def test_rays_discriminate_the_second_plane():
    for lift, least in ((S.BENT_LIFT_MESH, 3), (S.BENT_LIFT_FREE, 30)):
        q = S.bent_quad(-0.5, -0.5, lift)
        o, d = S.bent_quad_rays([q], seed=5)
        ok, _ = NP._quad(q.reshape(-1), o, d, np.full(len(o), np.inf))
        short = _same_plane_shortcut(q, o, d)
        robust = NP.quad_margin(q.reshape(-1), o, d) > 1e-6
        assert int(((ok != short) & robust).sum()) >= least
