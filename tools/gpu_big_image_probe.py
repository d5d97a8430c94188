"""Large images and large sample counts (device only): the Cornell box at 7680x4320, at 16384x16384 (the largest image
the library takes is 2^28 pixels) and at 2x2 pixels with 2^30 samples each, through both pipelines: exact sample counts,
equal ray counts, the mean radiance of a small render."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustraytracer_amd import api, config  # noqa: E402

flat = config.load_scene(os.path.join(ROOT, "configs/cornell_box/scene.toml"))
be, sc = api.Backend(0), api.Scene(flat)
small, _ = be.render_accumulate(sc, config.stratified_config(512, 512, 8, 8, seed=1).render_desc())
ref = small.mean() / 64
bad = 0
for w, h, side in ((7680, 4320, 2), (16384, 16384, 1), (16385, 3, 2), (2, 2, 32768)):
    rc = config.stratified_config(w, h, side, side, seed=2)
    rays = {}
    for name, pl in (("fused", api.PIPELINE_FUSED), ("wavefront", api.PIPELINE_WAVEFRONT)):
        img, st = be.render_accumulate(sc, rc.render_desc(pipeline=pl))
        mean = float(img.mean(dtype=np.float64)) / (side * side)
        ok = st["samples"] == w * h * side * side and st["nonfinite"] == 0 and (abs(mean - ref) < 0.03 * ref or h < 100)
        rays[name] = st["rays"]
        print("%dx%d @ %d %-9s samples %d rays %d mean %.4f (small render %.4f) kernel %.0f ms -> %s" %
              (w, h, side * side, name, st["samples"], st["rays"], mean, ref, st["kernel_ms"], "ok" if ok else "NOT ok"), flush=True)
        bad += 0 if ok else 1
        del img
    if abs(rays["fused"] - rays["wavefront"]) > 2e-6 * rays["fused"]:
        print("  ray counts differ:", rays)
        bad += 1
print("PROBLEMS", bad)
sys.exit(1 if bad else 0)
