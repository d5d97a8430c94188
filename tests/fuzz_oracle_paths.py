"""Not collected by pytest: a fuzz run of the f64 kernels against the oracle, PATH BY PATH, on random scenes
(python tests/fuzz_oracle_paths.py [scenes, default 100] [seed]).  The scenes of tools/gpu_exact_fuzz.py (every primitive
kind, scales 1e-3 .. 1e4, far offsets, slivers, near-duplicates, bent quads) with the five materials dealt out at random;
the oracle draws the device's numbers, so every pixel must agree to the f32 rounding of the stored sums -- fused kernel and
wavefront pipeline, scan and GPU-built tree, with and without next-event estimation."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import oracle_py as O  # noqa: E402
from rustraytracer_amd import _abi, api, config  # noqa: E402
import gpu_exact_fuzz as F  # noqa: E402


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    O.lib()
    be = api.Backend(0)
    total_off = total_px = worst = 0
    for s in range(n_scenes):
        flat, _, scale, _ = F.random_scene(rng)
        mats = ((_abi.MAT_EMISSIVE, (2.0, 1.5, 1.0), 0.0), (_abi.MAT_LAMBERTIAN, (0.7, 0.6, 0.5), 0.0), (_abi.MAT_REFLECTOR, (0, 0, 0), 0.0),
                (_abi.MAT_DIELECTRIC, (0, 0, 0), float(rng.choice([1.0, 1.33, 1.5, 2.4]))), (_abi.MAT_PHONG, (0.8, 0.7, 0.6), float(rng.choice([0.0, 1.0, 20.0, 500.0]))))
        m = np.zeros(len(mats), dtype=_abi.MATERIAL_DTYPE)
        for i, (typ, rgb, param) in enumerate(mats):
            m["type"][i], m["rgb"][i], m["param"][i] = typ, rgb, param
        flat.prims["object_id"] = rng.integers(0, len(mats), flat.n_prims)
        # next-event estimation refuses emitting planes: give those another material
        planes = (flat.prims["type"] == _abi.PRIM_PLANE) & (flat.prims["object_id"] == 0)
        flat.prims["object_id"][planes] = 1
        flat = config.FlatScene(flat.camera, flat.prims, m)
        sc = api.Scene(flat)
        rc = config.stratified_config(48, 36, 2, 2, min_depth=int(rng.integers(0, 3)), max_depth=int(rng.integers(1, 8)), seed=s) if s % 3 else \
            config.random_config(48, 36, 5, min_depth=1, max_depth=6, seed=s)
        W = 48 * 36
        for flags in (0, _abi.RENDER_NEE):
            ref, rst = O.render_accumulate(flat, rc.render_desc(flags=flags), device_numbers=True)
            for pl in (api.PIPELINE_FUSED, api.PIPELINE_WAVEFRONT):
                for accel in (api.ACCEL_LINEAR, api.ACCEL_SAH_GPU):
                    dev, st = be.render_accumulate(sc, rc.render_desc(precision=api.PRECISION_F64, flags=flags, pipeline=pl, accel=accel))
                    tol = (4e-7 if pl == api.PIPELINE_FUSED else 3e-7 * np.sqrt(2 * rc.spp + 2)) * np.abs(ref) + 1e-9
                    off = int(np.any(np.abs(dev.astype(np.float64) - ref) > tol, axis=-1).sum())
                    total_off += off
                    total_px += W
                    worst = max(worst, off)
                    if off > 6 or st["samples"] != rst["samples"] or st["nonfinite"] != rst["nonfinite"]:
                        print("scene %d flags %d pipeline %d accel %d: %d pixels off, rays %d / %d live, samples %d / %d, nonfinite %d / %d" %
                              (s, flags, pl, accel, off, st["rays"], rst["rays_live"], st["samples"], rst["samples"], st["nonfinite"], rst["nonfinite"]), flush=True)
        if s % 10 == 0:
            print("scene %3d (%3d primitives, scale %.1e, mean radiance %.3g, %d rays in the last render, lit pixels %d of %d): %d pixels off so far of %d" %
                  (s, flat.n_prims, scale, float(ref.mean()) / rc.spp, st["rays"], int(np.any(ref > 0, axis=-1).sum()), W, total_off, total_px), flush=True)
    print("TOTAL %d pixels off of %d (%.4f %%), worst render %d" % (total_off, total_px, 100.0 * total_off / max(total_px, 1), worst), flush=True)


if __name__ == "__main__":
    main()
