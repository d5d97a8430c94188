"""Independent numpy (f64) restatement of the reference's closest-hit query.

Written separately from oracle/oracle.c, vectorised over rays, looping over
primitives in scan order.  Used to cross-check the C oracle and the golden
primary-hit vectors.  Formulas follow the reference literally:
  sphere   src/shapes/sphere.rs:58-96       plane  src/shapes/plane.rs:25-44
  triangle src/shapes/triangle.rs:22-61     quad   src/shapes/quadrilateral.rs:56-117
  scan     src/scene.rs:35-47, src/shapes/mesh.rs:28-89
"""
import numpy as np

T_MIN = 0.001


def _dot(a, b):
    return np.sum(a * b, axis=-1)


def create_rays(cam, u, v):
    """Camera::create_ray (src/camera.rs:38-45) for arrays u, v."""
    o = np.asarray(cam["origin"], dtype=np.float64)
    p = (np.asarray(cam["lower_left_corner"]) + u[:, None] * np.asarray(cam["horizontal"])
         + v[:, None] * np.asarray(cam["vertical"]) - o)
    d = p * (1.0 / np.sqrt(_dot(p, p)))[:, None]
    return np.broadcast_to(o, d.shape).copy(), d


def _sphere(v, o, d, t_max):
    c, r = v[0:3], v[3]
    l = c - o
    t_ca = _dot(l, d)
    d2 = _dot(l, l) - t_ca * t_ca
    ok = ~((d2 < 0.0) | (d2 > r * r))
    with np.errstate(invalid="ignore"):
        t_hc = np.sqrt(r * r - d2)
    t0, t1 = t_ca - t_hc, t_ca + t_hc
    t0 = np.where(t0 < T_MIN, t1, t0)
    ok &= ~(t0 < T_MIN) & ~(t0 > t_max)
    return ok, t0


def _plane(v, o, d, t_max):
    point, n = v[0:3], v[3:6]
    den = _dot(d, n)
    ok = ~(np.abs(den) < 1e-6)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = _dot(point - o, n) / den
    ok &= ~(t < T_MIN) & ~(t > t_max)
    return ok, t


def _triangle(v, o, d, t_max):
    v0, v1, v2 = v[0:3], v[3:6], v[6:9]
    e1, e2 = v1 - v0, v2 - v0
    h = np.cross(d, e2)
    a = _dot(h, e1)
    ok = ~(np.abs(a) < 1e-6)
    with np.errstate(divide="ignore", invalid="ignore"):
        f = 1.0 / a
        s = o - v0
        u = f * _dot(s, h)
        ok &= ~((u < 0.0) | (u > 1.0))
        q = np.cross(s, e1)
        w = f * _dot(d, q)
        ok &= ~((w < 0.0) | (u + w > 1.0))
        t = f * _dot(q, e2)
    ok &= (t > T_MIN) & (t < t_max)
    return ok, t


def _quad(v, o, d, t_max):
    v0, v1, v2, v3 = v[0:3], v[3:6], v[6:9], v[9:12]
    e01, e03 = v1 - v0, v3 - v0
    p = np.cross(d, e03)
    det = _dot(p, e01)
    ok = ~(np.abs(det) < 1e-6)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        tv = o - v0
        alpha = _dot(tv, p) * inv
        ok &= ~(alpha < 0.0)
        q = np.cross(tv, e01)
        beta = _dot(d, q) * inv
        ok &= ~(beta < 0.0)
        second = (alpha + beta) > 1.0
        e23, e21 = v3 - v2, v1 - v2
        pp = np.cross(d, e21)
        detp = _dot(pp, e23)
        invp = 1.0 / detp
        tp = o - v2
        ap = _dot(tp, pp) * invp
        qp = np.cross(tp, e23)
        bp = _dot(d, qp) * invp
        ok2 = ~(np.abs(detp) < 1e-6) & ~(ap < 0.0) & ~(bp < 0.0)
        ok &= ~second | ok2
        t = _dot(q, e03) * inv
    ok &= ~(t < T_MIN) & ~(t > t_max)
    return ok, t


_TESTS = {0: _sphere, 1: _plane, 2: _triangle, 3: _quad}


def scene_intersect(prims, o, d):
    """-> (object_id, prim_id, t); misses are (-1, -1, inf)."""
    n = len(o)
    best_t = np.full(n, np.finfo(np.float64).max)
    oid = np.full(n, -1, dtype=np.int32)
    pid = np.full(n, -1, dtype=np.int32)
    for p in prims:
        ok, t = _TESTS[int(p["type"])](p["v"], o, d, best_t)
        best_t = np.where(ok, t, best_t)
        oid = np.where(ok, np.int32(p["object_id"]), oid)
        pid = np.where(ok, np.int32(p["prim_id"]), pid)
    t = np.where(oid >= 0, best_t, np.inf)
    return oid, pid, t


def quad_margin(v, o, d):
    """Distance of a ray from every decision the quad test takes (src/shapes/quadrilateral.rs:71-111): the smallest of
    |alpha|, |beta|, |alpha + beta - 1| and, where the second corner is consulted, |alpha'|, |beta'|.  A ray with a
    margin far above the rounding of the arithmetic gets the same verdict in any precision."""
    v0, v1, v2, v3 = v[0:3], v[3:6], v[6:9], v[9:12]
    e01, e03 = v1 - v0, v3 - v0
    p = np.cross(d, e03)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / _dot(p, e01)
        tv = o - v0
        alpha = _dot(tv, p) * inv
        beta = _dot(d, np.cross(tv, e01)) * inv
        e23, e21 = v3 - v2, v1 - v2
        pp = np.cross(d, e21)
        invp = 1.0 / _dot(pp, e23)
        tp = o - v2
        ap = _dot(tp, pp) * invp
        bp = _dot(d, np.cross(tp, e23)) * invp
    m = np.minimum(np.minimum(np.abs(alpha), np.abs(beta)), np.abs(alpha + beta - 1.0))
    second = (alpha >= 0) & (beta >= 0) & (alpha + beta > 1.0)
    m = np.where(second, np.minimum(m, np.minimum(np.abs(ap), np.abs(bp))), m)
    return np.where(np.isfinite(m), m, 0.0)
