#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the f64 CPU oracle (oracle/oracle.c).

primary_hits.npz   per scene: (u, v) of a 64x64 pixel-centre grid followed by
                   4096 seeded random points, and the oracle's (object_id,
                   prim_id, t) for the camera ray through each.  The ids follow
                   the reference's scan order: object_id = index of the
                   [[objects]] entry, prim_id = face index inside a mesh.
images.npz         per scene: 48x48 radiance means (float32, linear, before
                   post-processing) of two oracle renders with different seeds
                   at 32x32 = 1024 spp, used to calibrate the image-RMSE gate.

paths.npz          per scene: the pixel SUMS (float64) of 32x32 renders at 4x4 = 16 spp,
                   seed 61, with the oracle drawing the numbers of the HIP backend's
                   generator ("f64": what the f64 kernels must reproduce pixel by pixel;
                   "f32": what the f32 kernels follow up to their rounding), plus the
                   oracle's count of closest-hit queries made with non-zero throughput.

The reference itself cannot produce these (no Rust toolchain here, no
deterministic seed in the reference); the oracle is cross-checked against an
independent numpy restatement in tests/test_oracle_scene.py.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from rustraytracer_amd import config  # noqa: E402

SCENES = ("cornell_box", "smallpt")


def primary_uv():
    g = (np.arange(64) + 0.5) / 64.0
    gu, gv = np.meshgrid(g, g)
    rng = np.random.default_rng(20240607)
    u = np.concatenate([gu.ravel(), rng.random(4096)])
    v = np.concatenate([gv.ravel(), rng.random(4096)])
    return u, v


def main():
    out = {}
    img = {}
    u, v = primary_uv()
    for name in SCENES:
        flat = config.load_scene(os.path.join(ROOT, "configs", name, "scene.toml"))
        oid, pid, t = O.trace_primary(flat, u, v)
        out[f"{name}_u"], out[f"{name}_v"] = u, v
        out[f"{name}_object_id"], out[f"{name}_prim_id"], out[f"{name}_t"] = oid, pid, t
        for seed in (11, 12):
            rc = config.stratified_config(48, 48, 32, 32, seed=seed)
            rgb, _ = O.render_accumulate(flat, rc.render_desc())
            img[f"{name}_seed{seed}"] = (rgb / rc.spp).astype(np.float32)
    paths = {}
    for name in SCENES:
        flat = config.load_scene(os.path.join(ROOT, "configs", name, "scene.toml"))
        rc = config.stratified_config(32, 32, 4, 4, seed=61)
        for mode in ("f64", "f32"):
            rgb, st = O.render_accumulate(flat, rc.render_desc(), device_numbers=mode)
            paths[f"{name}_{mode}"] = rgb
            paths[f"{name}_{mode}_rays"] = np.int64(st["rays_live"])
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "paths.npz"), **paths)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "primary_hits.npz"), **out)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "images.npz"), **img)


if __name__ == "__main__":
    main()
