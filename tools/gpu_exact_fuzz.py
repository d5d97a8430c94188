"""Random scenes against the tracked f32 closest hit (device only): gpu_exact_fuzz.py [scenes, default 200] [seed]
Every scene: 5..300 primitives of all five kinds (spheres, planes, triangles, parallelograms, bent and unbent general
quads; slivers and near-duplicates among them) at a random scale (1e-3 .. 1e4) and a random offset from the origin (up to
1000 x the scale); 60 000 rays (between random points, and from points on the primitives into random directions, half of
them grazing).  For every ray the f32 answer (scan and GPU-built tree) must name the primitive the f64 answer names."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustraytracer_amd import _abi, api, config  # noqa: E402


def unit(rng, n):
    x = rng.normal(size=(n, 3))
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def random_scene(rng):
    n = int(rng.integers(5, 300))
    scale = 10.0 ** rng.uniform(-3, 4)
    offset = unit(rng, 1)[0] * scale * 10.0 ** rng.uniform(-1, 3) * (rng.random() < 0.7)
    p = np.zeros(n, dtype=_abi.PRIM_DTYPE)
    pts = []  # a point on (or near) every primitive, for the surface rays
    for i in range(n):
        kind = rng.choice([_abi.PRIM_SPHERE, _abi.PRIM_PLANE, _abi.PRIM_TRIANGLE, _abi.PRIM_QUAD, _abi.PRIM_QUAD], p=[0.12, 0.04, 0.4, 0.3, 0.14])
        c = offset + rng.normal(size=3) * scale
        if i and rng.random() < 0.08:  # nearly the place of the previous primitive
            c = pts[-1] + rng.normal(size=3) * scale * 1e-6
        size = scale * 10.0 ** rng.uniform(-2.5, 0.3)
        p["type"][i] = kind
        p["object_id"][i] = 0
        p["prim_id"][i] = i
        if kind == _abi.PRIM_SPHERE:
            p["v"][i, :4] = [*c, size]
            pts.append(c + unit(rng, 1)[0] * size)
        elif kind == _abi.PRIM_PLANE:
            nrm = unit(rng, 1)[0]
            p["v"][i, :6] = [*c, *nrm]
            t = np.cross(nrm, unit(rng, 1)[0])
            pts.append(c + t * scale * rng.uniform(0, 2))
        else:
            e1, e2 = unit(rng, 1)[0] * size, unit(rng, 1)[0] * size
            if rng.random() < 0.15:
                e2 = e1 * rng.uniform(0.2, 2.0) + e2 * 1e-4  # a sliver
            if kind == _abi.PRIM_TRIANGLE:
                v = np.array([c, c + e1, c + e2])
                p["v"][i, :9] = v.reshape(9)
                w = rng.dirichlet([1, 1, 1])
                pts.append(w @ v)
            else:
                bend = 0.0 if i % 3 else size * 10.0 ** rng.uniform(-9, -2.3)  # within the loader's 2.5 degrees, mostly far less
                lift = np.cross(e1, e2)
                lift = lift / (np.linalg.norm(lift) + 1e-300) * bend
                stretch = 1.0 if rng.random() < 0.5 else rng.uniform(0.6, 1.4)  # 1: a parallelogram
                v = np.array([c, c + e1, c + (e1 + e2) * stretch + lift, c + e2])
                p["v"][i, :12] = v.reshape(12)
                pts.append(c + e1 * rng.uniform(0, 1) * 0.5 + e2 * rng.uniform(0, 1) * 0.5)
    m = np.zeros(1, dtype=_abi.MATERIAL_DTYPE)
    m["type"], m["rgb"] = _abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5)
    cam = config.perspective_camera(tuple(offset + np.array([0, 0, 3 * scale])), tuple(offset), (0, 1, 0), 50.0, 1.0)
    return config.FlatScene(cam, p, m), np.array(pts), scale, offset


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    be = api.Backend(0)
    total = bad_total = 0
    for s in range(n_scenes):
        flat, pts, scale, offset = random_scene(rng)
        sc = api.Scene(flat)
        k = 30000
        a = offset + rng.normal(size=(k, 3)) * scale * 1.5
        b = pts[rng.integers(0, len(pts), k)] + rng.normal(size=(k, 3)) * scale * 10.0 ** rng.uniform(-6, -1)
        d1 = b - a
        d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
        o2 = pts[rng.integers(0, len(pts), k)]
        d2 = unit(rng, k)
        graze = rng.random(k) < 0.5
        ax = rng.integers(0, 3, k)
        d2[graze, ax[graze]] *= 1e-3
        d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
        o, d = np.concatenate([a, o2]), np.concatenate([d1, d2])
        bad_scene = 0
        for accel in (api.ACCEL_LINEAR, api.ACCEL_SAH_GPU):
            r64 = be.intersect(sc, o, d, precision=api.PRECISION_F64, accel=accel)
            r32 = be.intersect(sc, o, d, precision=api.PRECISION_F32, accel=accel)
            bad = (r64[0] != r32[0]) | (r64[1] != r32[1])
            # two primitives at the same f64 distance: either name is right (the tests treat exact ties the same way)
            with np.errstate(invalid="ignore"):
                tie = bad & (np.abs(r64[2] - r32[2]) <= 1e-12 * np.abs(r64[2]))
            bad_scene += int((bad & ~tie).sum())
            total += len(o)
        bad_total += bad_scene
        if bad_scene or s % 20 == 0:
            print("scene %3d: %3d primitives, scale %.1e, offset %.1e: %d rays named differently" %
                  (s, flat.n_prims, scale, float(np.linalg.norm(offset)), bad_scene), flush=True)
    print("TOTAL %d rays, %d named differently" % (total, bad_total), flush=True)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
