"""Larger scenes than the benchmark's (device only): gpu_scale_check.py [triangles ...]
For every size a displaced sphere of that many triangles (radius 100: triangle areas stay far above the reference's
absolute determinant threshold) inside an emissive sphere: the first-call time of every tree builder, the depth of the
tree, a small render through the wavefront pipeline, and the builders checked against each other -- the three trees
must name the same primitive and distance for every one of 200 000 rays (and, up to 4 M triangles, the scan too)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_mesh  # noqa: E402
from rustraytracer_amd import _abi, api, config  # noqa: E402


def scene(n_tri):
    v, _, f = make_mesh.displaced_sphere(n_tri, radius=100.0, bump=0.2, seed=3)
    tri = v[f].astype(np.float64) + np.array([0, 0, -400.0])
    p = np.zeros(len(f) + 1, dtype=_abi.PRIM_DTYPE)
    p["type"][: len(f)] = _abi.PRIM_TRIANGLE
    p["prim_id"][: len(f)] = np.arange(len(f))
    p["v"][: len(f), :9] = tri.reshape(len(f), 9)
    p["type"][len(f)], p["object_id"][len(f)] = _abi.PRIM_SPHERE, 1
    p["v"][len(f), :4] = [0, 0, 0, 4000.0]
    m = np.zeros(2, dtype=_abi.MATERIAL_DTYPE)
    m["type"][0], m["rgb"][0] = _abi.MAT_LAMBERTIAN, (0.7, 0.7, 0.7)
    m["type"][1], m["rgb"][1] = _abi.MAT_EMISSIVE, (1, 1, 1)
    cam = config.perspective_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 35.0, 1.0)
    return config.FlatScene(cam, p, m)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [4000000]
    be = api.Backend(0)
    rng = np.random.default_rng(4)
    n_rays = 200000
    for n in sizes:
        t0 = time.time()
        flat = scene(n)
        sc = api.Scene(flat)
        print("%d primitives (generated in %.1f s)" % (flat.n_prims, time.time() - t0), flush=True)
        # rays at the mesh from points around it
        target = np.array([0, 0, -400.0]) + rng.normal(size=(n_rays, 3)) * 60.0
        o = np.array([0, 0, -400.0]) + make_unit(rng, n_rays) * 250.0
        d = target - o
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rc = config.stratified_config(256, 256, 4, 4, seed=2)
        ref = None
        for name, accel in (("sah_gpu", api.ACCEL_SAH_GPU), ("lbvh", api.ACCEL_LBVH), ("sah", api.ACCEL_SAH)):
            _, st = be.render_accumulate(sc, rc.render_desc(accel=accel, pipeline=api.PIPELINE_WAVEFRONT))
            got = be.intersect(sc, o, d, accel=accel)
            same = "-" if ref is None else str(all(np.array_equal(a, b) for a, b in zip(ref, got)))
            if ref is None:
                ref = got
            print("  %-8s build %7.1f ms, depth %2d, %.2f node + %.2f primitive steps per ray, %.0f Mrays/s, stack overflows %d, hits %d of %d, same as sah_gpu: %s"
                  % (name, st["tree_build_ms"], st["tree_depth"], st["node_tests"] / st["rays"], st["prim_tests"] / st["rays"],
                     st["rays"] / st["kernel_ms"] / 1e3, st["stack_overflows"], int((got[0] == 0).sum()), n_rays, same), flush=True)
        if flat.n_prims <= 4100000:
            k = 20000
            lin = be.intersect(sc, o[:k], d[:k], accel=api.ACCEL_LINEAR)
            print("  scan of every primitive for %d rays: same as the trees: %s" % (k, all(np.array_equal(a, b[:k]) for a, b in zip(lin, ref))), flush=True)


def make_unit(rng, n):
    x = rng.normal(size=(n, 3))
    return x / np.linalg.norm(x, axis=1, keepdims=True)


if __name__ == "__main__":
    main()
