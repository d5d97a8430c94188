"""The tracked f32 closest hit against the f64 kernels, many rays (a soak run, device only):
   gpu_exact_probe.py [rays per class, default 4000000]
Three classes of rays per scene and structure -- camera rays at random (u, v); rays between random points of the scene's
box; rays that START ON A SURFACE (the f64 hit point of a camera ray, a random direction into the hemisphere: the class
that grazes its own surface and the neighbouring ones) -- and for each the number of rays whose f32 answer names another
primitive than the f64 answer, the share of f32 queries that were repeated in f64, and the renderer's speed with and
without RRT_RENDER_EXACT_HITS."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rustraytracer_amd import _abi, api, config  # noqa: E402


def scene_box(flat):
    """Box of the bounded primitives (without smallpt's wall spheres)."""
    P = flat.prims
    t, v = P["type"], P["v"]
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    sph = (t == _abi.PRIM_SPHERE) & (np.abs(v[:, 3]) < 1e3)
    if sph.any():
        lo = np.minimum(lo, (v[sph, :3] - np.abs(v[sph, 3:4])).min(axis=0))
        hi = np.maximum(hi, (v[sph, :3] + np.abs(v[sph, 3:4])).max(axis=0))
    for typ, nv in ((_abi.PRIM_TRIANGLE, 3), (_abi.PRIM_QUAD, 4)):
        m = t == typ
        if m.any():
            pts = v[m, : 3 * nv].reshape(-1, 3)
            lo, hi = np.minimum(lo, pts.min(axis=0)), np.maximum(hi, pts.max(axis=0))
    return lo, hi


def unit(rng, n):
    d = rng.normal(size=(n, 3))
    return d / np.linalg.norm(d, axis=1, keepdims=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
    be = api.Backend(0)
    scenes = {name: config.load_scene(os.path.join(ROOT, "configs", name, "scene.toml")) for name in ("cornell_box", "smallpt", "cornell_cube")}
    if os.path.exists(os.path.join(ROOT, "configs", "mesh_1m", "scene.toml")):
        import make_c5_assets
        make_c5_assets.main()
        scenes["mesh_1m"] = config.load_scene(os.path.join(ROOT, "configs", "mesh_1m", "scene.toml"))
    rng = np.random.default_rng(2026)
    total_bad = 0
    for name, flat in scenes.items():
        sc = api.Scene(flat)
        accels = [("sah_gpu", api.ACCEL_SAH_GPU), ("lbvh", api.ACCEL_LBVH)]
        if flat.n_prims < 1000:
            accels.insert(0, ("linear", api.ACCEL_LINEAR))
        u, v = rng.random(n), rng.random(n)
        lo, hi = scene_box(flat)
        a, b = lo + (hi - lo) * rng.random((n, 3)), lo + (hi - lo) * rng.random((n, 3))
        classes = {"camera": None, "box to box": (a, (b - a) / np.linalg.norm(b - a, axis=1, keepdims=True))}
        for label, accel in accels:
            t0 = time.time()
            p64 = be.trace_primary(sc, u, v, precision=api.PRECISION_F64, accel=accel)
            p32 = be.trace_primary(sc, u, v, precision=api.PRECISION_F32, accel=accel)
            bad = int(((p64[0] != p32[0]) | (p64[1] != p32[1])).sum())
            line = ["camera %d" % bad]
            total_bad += bad
            if "surface" not in classes:  # rays that start on the surfaces the camera sees
                cam = flat.camera  # Camera::create_ray (src/camera.rs:37-45)
                d = cam["lower_left_corner"] + u[:, None] * cam["horizontal"] + v[:, None] * cam["vertical"] - cam["origin"]
                d /= np.linalg.norm(d, axis=1, keepdims=True)
                hit = np.isfinite(p64[2]) & (p64[0] >= 0)
                o2 = (cam["origin"] + p64[2][:, None] * d)[hit]
                d2 = unit(rng, len(o2))
                # half of them nearly tangent to something: a small component along a random axis
                graze = rng.random(len(o2)) < 0.5
                axis = rng.integers(0, 3, len(o2))
                d2[graze, axis[graze]] *= 1e-3
                d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
                classes["surface"] = (o2, d2)
            for cls in ("box to box", "surface"):
                o, d = classes[cls]
                r64 = be.intersect(sc, o, d, precision=api.PRECISION_F64, accel=accel)
                r32 = be.intersect(sc, o, d, precision=api.PRECISION_F32, accel=accel)
                bad = int(((r64[0] != r32[0]) | (r64[1] != r32[1])).sum())
                total_bad += bad
                line.append("%s %d" % (cls, bad))
            print("%-12s %-7s %8d rays per class, f32 names another primitive than f64: %s  (%.1f s)" % (name, label, n, ", ".join(line), time.time() - t0), flush=True)
    print("TOTAL mismatches", total_bad, flush=True)
    rc = config.stratified_config(500, 500, 16, 16, seed=5)
    for name in ("cornell_box", "smallpt"):
        sc = api.Scene(scenes[name])
        for flags in (0, api.RENDER_EXACT_HITS):
            be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_FUSED, flags=flags))
            _, st = be.render_accumulate(sc, rc.render_desc(pipeline=api.PIPELINE_FUSED, flags=flags))
            print("render %-12s exact %d: %.2f ms, %.0f Msamples/s, queries repeated in f64 %.2e of %d" %
                  (name, flags != 0, st["kernel_ms"], st["samples"] / st["kernel_ms"] / 1e3, st["exact_redos"] / max(st["rays"], 1), st["rays"]), flush=True)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
