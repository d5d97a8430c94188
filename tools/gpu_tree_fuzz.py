"""Random scenes through the scan and through every tree (device only): gpu_tree_fuzz.py [scenes, default 100] [seed]
The scenes of gpu_exact_fuzz.py with four materials dealt out at random (an emitter among them).  A render must trace
the same paths whatever finds the closest hit: equal ray and bounce counts for the scan and the three trees, in both
pipelines and both precisions, and the same image to float summation order."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rustraytracer_amd import _abi, api, config  # noqa: E402
import gpu_exact_fuzz as F  # noqa: E402


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    be = api.Backend(0)
    bad_total = 0
    for s in range(n_scenes):
        flat, pts, scale, offset = F.random_scene(rng)
        m = np.zeros(4, dtype=_abi.MATERIAL_DTYPE)
        for i, (typ, rgb, param) in enumerate(((_abi.MAT_EMISSIVE, (2.0, 1.5, 1.0), 0.0), (_abi.MAT_LAMBERTIAN, (0.7, 0.6, 0.5), 0.0),
                                               (_abi.MAT_REFLECTOR, (0, 0, 0), 0.0), (_abi.MAT_DIELECTRIC, (0, 0, 0), 1.5))):
            m["type"][i], m["rgb"][i], m["param"][i] = typ, rgb, param
        flat.prims["object_id"] = rng.integers(0, 4, flat.n_prims)
        flat = config.FlatScene(flat.camera, flat.prims, m)
        sc = api.Scene(flat)
        rc = config.stratified_config(64, 48, 2, 2, min_depth=1, max_depth=6, seed=s)
        bad = 0
        for prec in (api.PRECISION_F32, api.PRECISION_F64):
            for pl in (api.PIPELINE_FUSED, api.PIPELINE_WAVEFRONT):
                ref = None
                for accel in (api.ACCEL_LINEAR, api.ACCEL_SAH_GPU, api.ACCEL_LBVH, api.ACCEL_SAH):
                    img, st = be.render_accumulate(sc, rc.render_desc(precision=prec, pipeline=pl, accel=accel))
                    key = (st["rays"], tuple(st["bounces"]), st["samples"])
                    if ref is None:
                        ref = (key, img)
                    elif key != ref[0] or not np.allclose(img, ref[1], rtol=2e-5, atol=1e-6) or st["stack_overflows"] or st["nonfinite"]:
                        bad += 1
                        print("scene %d precision %d pipeline %d accel %d: %s against the scan's %s" % (s, prec, pl, accel, key, ref[0]), flush=True)
        bad_total += bad
        if s % 20 == 0:
            print("scene %3d: %3d primitives, scale %.1e: %d differences" % (s, flat.n_prims, scale, bad), flush=True)
    print("TOTAL differences", bad_total, "in", n_scenes, "scenes", flush=True)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
