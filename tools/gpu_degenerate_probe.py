"""Degenerate inputs at sizes that reach the chunked path of the GPU tree builder (device only): many identical
triangles, many triangles around one centre, a long thin line of triangles, everything in one plane.  Every tree must
return the scan's hits for every ray."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustraytracer_amd import _abi, api, config  # noqa: E402


def scene_of(tri):
    n = len(tri)
    p = np.zeros(n, dtype=_abi.PRIM_DTYPE)
    p["type"] = _abi.PRIM_TRIANGLE
    p["prim_id"] = np.arange(n)
    p["v"][:, :9] = tri.reshape(n, 9)
    m = np.zeros(1, dtype=_abi.MATERIAL_DTYPE)
    m["type"], m["rgb"] = _abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5)
    return config.FlatScene(config.perspective_camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 50.0, 1.0), p, m)


def main():
    rng = np.random.default_rng(1)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    base = np.array([[-1.0, -1.0, 0.0], [1.0, -1.0, 0.0], [0.0, 1.0, 0.0]])
    cases = {}
    cases["identical"] = np.repeat(base[None], n, axis=0)
    cases["same centre"] = base[None] * rng.uniform(0.01, 1.0, size=(n, 1, 1))
    line = base[None] * 1e-3 + np.stack([np.linspace(-50, 50, n), np.zeros(n), np.zeros(n)], axis=1)[:, None, :]
    cases["line"] = line
    flat = base[None] * 0.01 + np.concatenate([rng.uniform(-3, 3, size=(n, 1, 2)), np.zeros((n, 1, 1))], axis=2)
    cases["one plane"] = flat
    two = np.concatenate([np.repeat(base[None], n // 2, axis=0), np.repeat((base + [5, 0, 0])[None], n - n // 2, axis=0)])
    cases["two piles"] = two
    be = api.Backend(0)
    bad = 0
    for name, tri in cases.items():
        flat_scene = scene_of(tri)
        sc = api.Scene(flat_scene)
        k = 3000
        target = tri[rng.integers(0, len(tri), k)].mean(axis=1) + rng.normal(size=(k, 3)) * 0.01
        o = target + np.array([0, 0, 4.0]) + rng.normal(size=(k, 3)) * 0.5
        d = target - o
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        lin = be.intersect(sc, o, d, accel=api.ACCEL_LINEAR)
        for label, accel in (("sah_gpu", api.ACCEL_SAH_GPU), ("lbvh", api.ACCEL_LBVH), ("sah", api.ACCEL_SAH)):
            t0 = time.time()
            got = be.intersect(sc, o, d, accel=accel)
            same = all(np.array_equal(a, b) for a, b in zip(lin, got))
            _, st = be.render_accumulate(sc, config.stratified_config(64, 64, 2, 2).render_desc(accel=accel))
            print("%-12s %-8s %d triangles: same as the scan %s, hits %d, depth %d, build %.0f ms, first call %.2f s, stack overflows %d"
                  % (name, label, len(tri), same, int((lin[0] >= 0).sum()), st["tree_depth"], st["tree_build_ms"], time.time() - t0, st["stack_overflows"]), flush=True)
            bad += 0 if same and st["stack_overflows"] == 0 else 1
    print("PROBLEMS", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
