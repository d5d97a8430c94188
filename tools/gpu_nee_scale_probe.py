"""The two pipelines against each other on the million-triangle scene (device only), with and without next-event
estimation (shadow rays through the shadow queue and the persistent tree kernel): they must agree ray for ray and pixel
for pixel (to float summation order) in both precisions.  (In f32 that holds because the library is built with
-ffp-contract=on: every kernel then rounds the same expressions the same way.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_c5_assets  # noqa: E402
from rustraytracer_amd import _abi, api, config  # noqa: E402

make_c5_assets.main()
flat = config.load_scene(os.path.join(ROOT, "configs/mesh_1m/scene.toml"))
be, sc = api.Backend(0), api.Scene(flat)
bad = 0
for prec, w, h, side, flags in ((api.PRECISION_F32, 480, 270, 4, 0), (api.PRECISION_F32, 480, 270, 4, _abi.RENDER_NEE), (api.PRECISION_F64, 160, 90, 2, _abi.RENDER_NEE)):
    rc = config.stratified_config(w, h, side, side, seed=3)
    out = {}
    print("flags", flags)
    for name, pl in (("fused", api.PIPELINE_FUSED), ("wavefront", api.PIPELINE_WAVEFRONT)):
        img, st = be.render_accumulate(sc, rc.render_desc(precision=prec, flags=flags, pipeline=pl))
        out[name] = (img.astype(np.float64), st)
        print("precision %d %-9s rays %d samples %d nonfinite %d stack overflows %d kernel %.1f ms accel %d" %
              (prec, name, st["rays"], st["samples"], st["nonfinite"], st["stack_overflows"], st["kernel_ms"], st["accel_used"]), flush=True)
    a, b = out["fused"], out["wavefront"]
    rel = np.abs(a[0] - b[0]) / (np.abs(a[0]) + 1e-3)
    above = int(np.any(rel > 1e-4, axis=-1).sum())
    rays_rel = abs(a[1]["rays"] - b[1]["rays"]) / a[1]["rays"]
    ok = rays_rel == 0 and int(np.any(rel > (2e-6 if prec == api.PRECISION_F64 else 2e-5), axis=-1).sum()) == 0
    ok = ok and a[1]["nonfinite"] == 0 and b[1]["nonfinite"] == 0 and a[1]["stack_overflows"] == 0 and b[1]["stack_overflows"] == 0
    print("  ray counts differ by %.1e, pixels above 1e-4: %d of %d -> %s" % (rays_rel, above, w * h, "ok" if ok else "NOT ok"), flush=True)
    bad += 0 if ok else 1
print("PROBLEMS", bad)
sys.exit(1 if bad else 0)
